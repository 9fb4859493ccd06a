// Instantiates the batched AL-iLQR engine for (float, UnicycleM) on gfx950.
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineUnicycleF32(const altro_desc& d, std::string* err) { return MakeEngineImpl<float, UnicycleM>(d, err); }
}  // namespace altro_hip
