// Device pieces of SciPy's order-3 B-spline resampling (scipy/ndimage/src/ni_splines.c, ni_interpolation.c), shared by
// zoom.hip (ndimage.zoom inside common.process_samples) and augment.hip (ndimage.rotate / clipped zoom of train.DataGenerator).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace rml_spline {

__device__ inline void prefilter_line(double* c, int n, int stride) {
    if (n < 2) return;
    const double z = -0.26794919243112270647;     // sqrt(3) - 2
    const double lam = (1.0 - z) * (1.0 - 1.0 / z);
    for (int i = 0; i < n; ++i) c[i * stride] *= lam;
    const double zn1 = pow(z, (double)(n - 1));
    double c0 = c[0] + zn1 * c[(n - 1) * stride];
    double zi = z, z2 = zn1 * zn1 / z;
    for (int i = 1; i < n - 1; ++i) {
        c0 += (zi + z2) * c[i * stride];
        zi *= z; z2 /= z;
    }
    c[0] = c0 / (1.0 - zn1 * zn1);
    for (int i = 1; i < n; ++i) c[i * stride] += z * c[(i - 1) * stride];
    c[(n - 1) * stride] = (z * c[(n - 2) * stride] + c[(n - 1) * stride]) * z / (z * z - 1.0);
    for (int i = n - 2; i >= 0; --i) c[i * stride] = z * (c[(i + 1) * stride] - c[i * stride]);
}

__device__ __forceinline__ int mirror_idx(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i = i < 0 ? -i : i;
    i %= p;
    return i >= n ? p - i : i;
}

__device__ __forceinline__ void bspline3(double t, double w[4]) {
    const double t2 = t * t, t3 = t2 * t, u = 1.0 - t;
    w[0] = u * u * u / 6.0;
    w[1] = (3.0 * t3 - 6.0 * t2 + 4.0) / 6.0;
    w[2] = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) / 6.0;
    w[3] = t3 / 6.0;
}

}  // namespace rml_spline
