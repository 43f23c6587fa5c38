// C-ABI wrapper around the REFERENCE's own extract_context_portions, compiled from the sources
// where they lie under /root/reference (see oracle/Makefile, target _ref).  Test infrastructure
// only: it lets tests compare oracle/pnn_oracle.c (and through it the HIP gather) with the
// reference function itself.  No reference source is copied into this repository.
#include "extraction_context.h"  // resolved with -I/root/reference/hevc/hm_common/c++/source_common
#include <cstdint>
#include <vector>

extern "C" int ref_extract_context_portions(const int32_t* roi_origin, float* above, float* left,
                                            const uint8_t* flags, int n_flags, int n_avail,
                                            int unit_w, int unit_h, int above_units, int left_units,
                                            int tu_w, int tu_h, int stride, float mean)
{
    // The reference takes `const bool*`; copy the byte flags into real bools.
    std::vector<char> tmp(n_flags > 0 ? n_flags : 1);
    bool* b = reinterpret_cast<bool*>(tmp.data());
    for (int i = 0; i < n_flags; i++) b[i] = flags[i] != 0;
    return extract_context_portions(roi_origin, above, left, b, n_avail, unit_w, unit_h,
                                    above_units, left_units, tu_w, tu_h, stride, mean);
}
