// Host-only part of the C ABI (no HIP in this file; tests/sanitize_host.cpp runs it under ASan / UBSan): the batch-1
// context gather that HM calls from TComPrediction::initIntraPatternChType (TComPattern.cpp:367-380), the translation of
// HM's neighbour flags into a device descriptor, and the model-table parser.  Same contract as the reference's
// extract_context_portions (hevc/hm_common/c++/source_common/extraction_context.cpp:3-208): same
// argument order, -1 + a line on stderr for NULL pointers, a non-positive neighbour count or an
// unavailable corner unit.  Written around the same (above_mask, left_units) descriptor the GPU
// gather consumes, so both paths share one definition of "available".
#include "../../include/pnn_hip.h"

#include "pnn_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace pnn {

bool read_file(const std::string& path, std::vector<char>* out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(sz > 0 ? sz : 0);
    const size_t got = sz > 0 ? fread(out->data(), 1, sz, f) : 0;
    fclose(f);
    return got == (size_t)std::max(sz, 0L);
}

// hevc/hm_common/c++/source_common/tools.cpp:52-111 (+ split_string :127-152): fields split on runs of
// delimiters, lines made of whitespace only are skipped, keys parsed like std::stoul (leading blanks
// skipped, trailing text ignored), the path trimmed of surrounding whitespace.
int parse_table(const char* path, std::vector<TableEntry>* out, std::string* err)
{
    std::vector<char> data;
    if (!path || !read_file(path, &data)) { *err = std::string("The file at \"") + (path ? path : "(null)") + "\" cannot be opened."; return PNN_E_IO; }
    const std::string text(data.begin(), data.end());
    size_t pos = 0;
    const char* ws = " \t\f\v\n\r";
    while (pos <= text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        if (line.find_first_not_of(ws) == std::string::npos) { if (eol == text.size()) break; continue; }
        std::vector<std::string> f;
        size_t i = 0;
        while (i <= line.size()) {
            size_t j = line.find_first_of(",;", i);
            if (j == std::string::npos) { f.push_back(line.substr(i)); break; }
            f.push_back(line.substr(i, j - i));
            i = line.find_first_not_of(",;", j);
            if (i == std::string::npos) break;
        }
        if (f.size() < 4) { *err = "model table line with fewer than 4 fields: " + line; return PNN_E_IO; }
        TableEntry e;
        char* endp = nullptr;
        const char* s0 = f[0].c_str();
        e.width = (int)strtoul(s0, &endp, 10);
        if (endp == s0) { *err = "model table: bad width in line: " + line; return PNN_E_IO; }
        const char* s1 = f[1].c_str();
        e.is_pair = strtoul(s1, &endp, 10) != 0;
        if (endp == s1) { *err = "model table: bad is_pair in line: " + line; return PNN_E_IO; }
        const char* s2 = f[2].c_str();
        e.channel = (int)strtoul(s2, &endp, 10);
        if (endp == s2) { *err = "model table: bad channel in line: " + line; return PNN_E_IO; }
        std::string v = f[3];
        const size_t a = v.find_first_not_of(ws);
        const size_t b = v.find_last_not_of(ws);
        e.path = a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
        out->push_back(e);
        if (eol == text.size()) break;
    }
    return PNN_OK;
}

}  // namespace pnn

using namespace pnn;

extern "C" {

int pnn_make_tb_desc(pnn_tb_dev* out, int64_t origin, int32_t stride, const uint8_t* flags, int n_avail, int above_units,
                     int left_units)
{
    if (!out || !flags) { fprintf(stderr, "`out` or `neighbor_flags` is NULL.\n"); return -1; }
    if (n_avail <= 0) { fprintf(stderr, "`iNumIntraNeighbor` is not strictly positive.\n"); return -1; }   // extraction_context.cpp:42-47
    if (above_units > 32 || left_units < 0 || above_units < 0) return -1;
    out->origin = origin; out->stride = stride; out->reserved = 0;
    if (n_avail == above_units + left_units + 1) {               // extraction_context.cpp:56: dense copy of everything
        out->above_mask = above_units >= 32 ? 0xffffffffu : ((1u << above_units) - 1u);
        out->left_units = left_units;
        return 0;
    }
    if (!flags[left_units]) {                                     // extraction_context.cpp:133-139
        fprintf(stderr, "The neighbouring unit above and on the left side of the current TB is not available.\n");
        return -1;
    }
    uint32_t mask = 0;
    for (int i = 0; i < above_units; i++) if (flags[left_units + 1 + i]) mask |= 1u << i;
    int cnt = 0;
    for (int i = 0; i < left_units; i++) cnt += flags[i] != 0;   // rows compact upwards, extraction_context.cpp:189-205
    out->above_mask = mask; out->left_units = cnt;
    return 0;
}

int pnn_parse_model_table(const char* path, int* widths, int* is_pair, int* channels, const char** paths, int max_entries)
{
    static thread_local std::vector<TableEntry> keep;
    keep.clear();
    std::string err;
    const int rc = parse_table(path, &keep, &err);
    if (rc) { set_create_error(err); fprintf(stderr, "%s\n", err.c_str()); return rc; }
    const int n = std::min((int)keep.size(), max_entries);
    for (int i = 0; i < n; i++) {
        if (widths) widths[i] = keep[i].width;
        if (is_pair) is_pair[i] = keep[i].is_pair;
        if (channels) channels[i] = keep[i].channel;
        if (paths) paths[i] = keep[i].path.c_str();
    }
    return n;
}

}  // extern "C"

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
