// Stub (test infrastructure only): forward declarations come with the stand-in itself.
#include "open_spiel/json/include/nlohmann/json.hpp"
