// include-path shim: see open_spiel/spiel.h in this directory
#include "open_spiel/spiel.h"
