// ge_cost.h -- per-nonzero constants of the cost functions, shared by the update kernels (glove.hip) and the layout
// builder (glove_layout.hip).  Compile with -ffp-contract=off.
#pragma once
#include "ge_common.h"

// ---- per-nonzero constants of the cost functions -------------------------------------
// GloveCost:  ic = s + (fB+cB) - log(X);  wc = (X > max) ? ic : (float)pow(X/max, 0.75) * ic
// PGloveCost: ic = s + (fB+cB) - log(X/(1-X)) [fp32 division];  wc = X * ic
// Both reduce to  ic = (float)((double)s + ((double)(fB+cB) - l)),  wc = w * ic.
// EXACT = the deterministic kernel (libm pow, as FastMath.pow in GloveCost.java:19); the Hogwild
// kernel forms r^0.75 as sqrt(r)*sqrt(sqrt(r)) in fp64 (<= 2 ulp of fp64 before the fp32
// narrowing, far inside its tolerance) because pow() alone costs ~60 VGPRs of occupancy.
template <bool EXACT>
__device__ __forceinline__ void cost_terms(int kind, float x, double xmax, double &l, float &w) {
    if (kind == GE_COST_GLOVE) {
        l = log((double)x);
        const double r = (double)x / xmax;
        if (EXACT) w = ((double)x > xmax) ? 1.0f : (float)pow(r, 0.75);
        else { const double q = sqrt(r); w = ((double)x > xmax) ? 1.0f : (float)(q * sqrt(q)); }
    } else {
        l = log((double)(x / (1.0f - x)));
        w = x;
    }
}

