// Stub of the private abseil stand-in (test infrastructure only): see ../shim_all.h.
#include "open_spiel/abseil-cpp/absl/shim_all.h"
