// Instantiates the batched AL-iLQR engine for (double, WithRec32<UnicycleM>): the ALTRO_F32 dtype = fp32 expansion/gain records, fp64 state and arithmetic on gfx950.
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineUnicycleF32(const altro_desc& d, std::string* err) { return MakeEngineImpl<double, WithRec32<UnicycleM>>(d, err); }
}  // namespace altro_hip
