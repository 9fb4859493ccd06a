// Forwarding header: keeps the reference's include path `examples/triple_integrator.hpp` valid for callers that switch to the
// MI355X solver; everything lives in the one facade header.
#pragma once
#include "../altro/altro.hpp"
