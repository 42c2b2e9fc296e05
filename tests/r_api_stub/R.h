/* TEST-ONLY stand-in for <R.h> (see Rinternals.h in this directory) */
#ifndef PLAIDHIP_TEST_R_H
#define PLAIDHIP_TEST_R_H
#include <stdlib.h>
#endif
