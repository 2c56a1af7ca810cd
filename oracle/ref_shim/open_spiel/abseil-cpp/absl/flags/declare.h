// Stub of the private abseil stand-in (test infrastructure only): see shim_flags.h.
#include "open_spiel/abseil-cpp/absl/flags/shim_flags.h"
