// batched_problem.hpp — device-side description of a batch of same-shape LPs.
#pragma once

#include "lp_internal.hpp"

struct BatchedDev {
    int batch, m, n;
    int pitch;        // row pitch (doubles) of the condensed LDS tableau, odd
    int maximize;
    int max_iter;
    double eps;
    const double* A;        // batch x (m*n), each column-major
    const double* b;        // batch x m
    const double* c;        // batch x n
    const int* basis_in;    // batch x m
    double* x;              // batch x n  (full vertex)
    int* basis_out;         // batch x m  (by position)
    int* iters;             // batch
    int* status;            // batch
    unsigned long long* stamps;   // diagnostic (LP_BATCHED_STAMPS): 32 per-phase cycle sums of workgroup 0; nullptr = off
    int stamps_reg;               // LP_BATCHED_STAMPS=reg: the register form's instrumented instantiation (else the LDS form's)
    int pad_stamps;
};

// batched_simplex.hip
size_t lp_batched_lds_bytes(int m, int n, int* pitch_out);
int lp_batched_launch(lp_context* ctx, const BatchedDev& d);
