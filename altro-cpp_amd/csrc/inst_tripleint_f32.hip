// Instantiates the batched AL-iLQR engine for (float, TripleIntegratorM<2>) on gfx950.
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineTripleInt2F32(const altro_desc& d, std::string* err) { return MakeEngineImpl<float, TripleIntegratorM<2>>(d, err); }
}  // namespace altro_hip
