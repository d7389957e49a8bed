// Host-side tables of Pillow's bicubic resize (src/libImaging/Resample.c: precompute_coeffs with the bicubic filter, a = -0.5),
// shared by the bit-identical resize (resize.hip) and the float32 preprocessing kernel of the CNN path (preprocess.hip).
#pragma once
#include <math.h>
#include <vector>

namespace rmlresize {

struct AxisTable {          // precompute_coeffs() for one (in, out) pair
    int in = 0, out = 0, ksize = 0;
    std::vector<int> bounds;        // [out][2]: first input index, tap count
    std::vector<double> kk;         // [out][ksize]: normalised weights
};

inline double bicubic_filter(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

inline void precompute(int in_size, int out_size, AxisTable& t) {
#pragma clang fp contract(off)
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    t.in = in_size; t.out = out_size;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * t.ksize, 0.0);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double* k = &t.kk[(size_t)xx * t.ksize];
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        t.bounds[2 * xx] = xmin;
        t.bounds[2 * xx + 1] = xmax;
    }
}

}  // namespace rmlresize
