// Prelude for compiling OpenCV-free SLICES of the reference in place (oracle/Makefile).
// Nothing from /root/reference is copied into the repo: the Makefile pipes line ranges of
// the reference files straight into the compiler between this prelude and the epilogue.
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>
#ifndef CV_PI
#define CV_PI 3.1415926535897932384626433832795
#endif
