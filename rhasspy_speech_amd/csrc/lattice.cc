// (filled in below) host-side lattice determinisation / n-best
#include "lattice.h"
