// C-ABI wrapper around the REFERENCE's own distortion functions (TComRdCost::setDistParam + DistFunc, the pair
// TEncSearch.cpp:2300-2389 uses for the first intra pass), compiled from the sources where they lie under
// /root/reference (see oracle/Makefile, target ref).  Test infrastructure only; no reference source is copied.
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComRom.h"
#include <cstdint>

extern "C" unsigned ref_block_cost(const int32_t* org, int org_stride, const int32_t* cur, int cur_stride, int w, int h, int hadamard)
{
    static bool rom = (initROM(), true);   // g_aucConvertToBit selects the per-width function
    (void)rom;
    TComRdCost rd;
    DistParam dp;
    rd.setDistParam(dp, 8, org, org_stride, cur, cur_stride, w, h, hadamard != 0);
    return (unsigned)dp.DistFunc(&dp);
}
