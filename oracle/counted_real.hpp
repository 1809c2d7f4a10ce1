// oracle/counted_real.hpp — arithmetic type of the FLOP-COUNTING build of the CPU oracle (test / measurement infrastructure only).
//
// `make -C oracle count` compiles the very same tmjx_oracle.c / tmjx_oracle_env.c as C++ with `real` = this class: every + - * /
// on a `real` bumps a thread-local counter, as do the R_SQRT / R_SIN / ... wrappers (one "flop" each) and min / max / abs (counted
// separately: vector-ALU instructions on the GPU, not floating-point operations).  bench.py freezes the numbers it reports from
// oracle/flop_count.json, produced by oracle/count_flops.py with this build (the instrumented count that
// replaces the 1.1 Mflop/substep estimate).
#pragma once
#include <math.h>

struct OCount { unsigned long long add, mul, div, special, minmax; };
extern thread_local OCount g_ocount;

struct real {
  double v;
  real() = default;
  real(double x) : v(x) {}
  real(float x) : v(x) {}
  real(int x) : v(x) {}
  real(long x) : v((double)x) {}
  real(unsigned x) : v(x) {}
  operator double() const { return v; }
  real &operator+=(real o) { g_ocount.add++; v += o.v; return *this; }
  real &operator-=(real o) { g_ocount.add++; v -= o.v; return *this; }
  real &operator*=(real o) { g_ocount.mul++; v *= o.v; return *this; }
  real &operator/=(real o) { g_ocount.div++; v /= o.v; return *this; }
  real operator-() const { return real(-v); }
};
#define OC_BIN(op, ctr)                                                                               \
  inline real operator op(real a, real b) { g_ocount.ctr++; return real(a.v op b.v); }               \
  inline real operator op(real a, double b) { g_ocount.ctr++; return real(a.v op b); }               \
  inline real operator op(double a, real b) { g_ocount.ctr++; return real(a op b.v); }               \
  inline real operator op(real a, float b) { g_ocount.ctr++; return real(a.v op (double)b); }        \
  inline real operator op(float a, real b) { g_ocount.ctr++; return real((double)a op b.v); }        \
  inline real operator op(real a, int b) { g_ocount.ctr++; return real(a.v op (double)b); }          \
  inline real operator op(int a, real b) { g_ocount.ctr++; return real((double)a op b.v); }
OC_BIN(+, add)
OC_BIN(-, add)
OC_BIN(*, mul)
OC_BIN(/, div)
#undef OC_BIN
#define OC_CMP(op)                                                         \
  inline bool operator op(real a, real b) { return a.v op b.v; }           \
  inline bool operator op(real a, double b) { return a.v op b; }           \
  inline bool operator op(double a, real b) { return a op b.v; }           \
  inline bool operator op(real a, int b) { return a.v op (double)b; }      \
  inline bool operator op(int a, real b) { return (double)a op b.v; }      \
  inline bool operator op(real a, float b) { return a.v op (double)b; }    \
  inline bool operator op(float a, real b) { return (double)a op b.v; }
OC_CMP(<) OC_CMP(>) OC_CMP(<=) OC_CMP(>=) OC_CMP(==) OC_CMP(!=)
#undef OC_CMP
inline real oc_special(double r) { g_ocount.special++; return real(r); }
inline real oc_minmax(double r) { g_ocount.minmax++; return real(r); }
#define R_SQRT(x) oc_special(sqrt((double)(x)))
#define R_SIN(x) oc_special(sin((double)(x)))
#define R_COS(x) oc_special(cos((double)(x)))
#define R_EXP(x) oc_special(exp((double)(x)))
#define R_ACOS(x) oc_special(acos((double)(x)))
#define R_POW(x, y) oc_special(pow((double)(x), (double)(y)))
#define R_FLOOR(x) oc_minmax(floor((double)(x)))
#define R_FABS(x) oc_minmax(fabs((double)(x)))
#define R_FMAX(x, y) oc_minmax(fmax((double)(x), (double)(y)))
#define R_FMIN(x, y) oc_minmax(fmin((double)(x), (double)(y)))
#define R_MAX 1.7976931348623157e308
