// Instantiates the ALTRO_F32 engine of the 2-dof triple integrator on gfx950: fp64 state and arithmetic,
// the expansion and gain records stored in fp32 (WithRec32<>, see altro_device.hpp).
#include "altro_engine.hpp"
namespace altro_hip {
EngineBase* MakeEngineTripleInt2F32(const altro_desc& d, std::string* err) { return MakeEngineImpl<double, WithRec32<TripleIntegratorM<2>>>(d, err); }
}  // namespace altro_hip
