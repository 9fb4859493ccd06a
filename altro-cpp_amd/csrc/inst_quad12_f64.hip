// Instantiates the batched AL-iLQR engine for (double, Quadrotor12M) on gfx950.
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineQuad12F64(const altro_desc& d, std::string* err) { return MakeEngineImpl<double, Quadrotor12M>(d, err); }
}  // namespace altro_hip
