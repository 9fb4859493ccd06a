// Instantiates the batched AL-iLQR engine for (double, TripleIntegratorM<2>) on gfx950.
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineTripleInt2F64(const altro_desc& d, std::string* err) { return MakeEngineImpl<double, TripleIntegratorM<2>>(d, err); }
}  // namespace altro_hip
