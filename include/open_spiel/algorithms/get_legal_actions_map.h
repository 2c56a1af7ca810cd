// forwards to the MI355X host mirror: see include/open_spiel/spiel.h
#include "open_spiel/spiel.h"
