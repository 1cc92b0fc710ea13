/*
 * clrs_oracle_mp.cpp -- CPU ORACLE at the reference's working precision (test infrastructure, NOT the product).
 *
 * Compiles the restatement in clrs_oracle.c a third time, as C++, with REAL = mpx<ORACLE_MP> (mpx.hpp): a binary
 * multi-precision float of ORACLE_MP 64-bit limbs, truncated to oracle_set_precision_bits() bits per operation.
 * It stands in for the reference's Arb midpoints at `prec` = 256 bits (src/solver.jl:73,103; src/tools.jl:59-107):
 * libclrs_oracle_mp.so (5 limbs = 320 bits, so that 256- and 300-bit runs of test/runtests_solver.jl:19-22 are
 * both covered) is the checker of the multi-word HIP path and the "reference precision" leg of bench.py's cpu_baseline.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "mpx.hpp"
#ifndef ORACLE_MP
#define ORACLE_MP 5
#endif
extern "C" {
#include "clrs_oracle.c"
}
