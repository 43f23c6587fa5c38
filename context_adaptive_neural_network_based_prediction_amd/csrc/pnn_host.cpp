// Host-only part of the C ABI: the batch-1 context gather that HM calls from
// TComPrediction::initIntraPatternChType (TComPattern.cpp:367-380).  Same contract as the reference's
// extract_context_portions (hevc/hm_common/c++/source_common/extraction_context.cpp:3-208): same
// argument order, -1 + a line on stderr for NULL pointers, a non-positive neighbour count or an
// unavailable corner unit.  Written around the same (above_mask, left_units) descriptor the GPU
// gather consumes, so both paths share one definition of "available".
#include "../../include/pnn_hip.h"

#include <cstdio>

extern "C" int pnn_extract_context(const int32_t* roi_origin, float* above, float* left, const uint8_t* neighbor_flags,
                                   int n_avail, int unit_w, int unit_h, int above_units, int left_units, int tu_w,
                                   int tu_h, int pic_stride, float mean)
{
    if (!roi_origin) { fprintf(stderr, "`piRoiOrigin` is NULL.\n"); return -1; }
    if (!above) { fprintf(stderr, "`piPortionAbove` is NULL.\n"); return -1; }
    if (!left) { fprintf(stderr, "`piPortionLeft` is NULL.\n"); return -1; }
    if (!neighbor_flags) { fprintf(stderr, "`bNeighborFlags` is NULL.\n"); return -1; }
    if (n_avail <= 0) { fprintf(stderr, "`iNumIntraNeighbor` is not strictly positive.\n"); return -1; }

    const bool all = n_avail == above_units + left_units + 1;
    if (!all && !neighbor_flags[left_units]) {
        fprintf(stderr, "The neighbouring unit above and on the left side of the current TB is not available.\n");
        return -1;
    }
    const int ctx_w = 3 * tu_w;

    // Above portion: rows [y - h, y), columns [x - w, x + 2w); corner block always, one strip per unit.
    const int32_t* src = roi_origin - (long)tu_h * pic_stride - tu_w;
    for (int r = 0; r < tu_h; r++, src += pic_stride) {
        float* dst = above + (long)r * ctx_w;
        for (int col = 0; col < tu_w; col++) dst[col] = (float)src[col] - mean;
        for (int col = tu_w; col < ctx_w; col++) {
            const int u = (col - tu_w) / unit_w;
            const bool ok = all || (u < above_units && neighbor_flags[left_units + 1 + u]);
            dst[col] = ok ? (float)src[col] - mean : 0.f;
        }
    }

    // Left portion: the first unit_h * (number of available left units) rows below y, then zeros
    // (source and destination advance together, only on available units).
    int rows = 2 * tu_h;
    if (!all) {
        int cnt = 0;
        for (int i = 0; i < left_units; i++) cnt += neighbor_flags[i] != 0;
        rows = cnt * unit_h;
    }
    src = roi_origin - tu_w;
    for (int r = 0; r < 2 * tu_h; r++, src += pic_stride) {
        float* dst = left + (long)r * tu_w;
        if (r < rows)
            for (int col = 0; col < tu_w; col++) dst[col] = (float)src[col] - mean;
        else
            for (int col = 0; col < tu_w; col++) dst[col] = 0.f;
    }
    return 0;
}
