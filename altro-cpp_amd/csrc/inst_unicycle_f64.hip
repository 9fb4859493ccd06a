// Instantiates the batched AL-iLQR engine for (double, UnicycleM) on gfx950.
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineUnicycleF64(const altro_desc& d, std::string* err) { return MakeEngineImpl<double, UnicycleM>(d, err); }
}  // namespace altro_hip
