// C-ABI wrapper around the REFERENCE's own model-table parser (parse_file_strings_three_keys,
// hevc/hm_common/c++/source_common/tools.cpp:52-111), compiled from the sources where they lie under
// /root/reference (oracle/Makefile, target ref).  Test infrastructure only: tests compare pnn_parse_model_table
// (csrc/pnn_host.cpp) with it on fuzzed tables.  No reference source is copied into this repository.
#include "tools.h"  // resolved with -I/root/reference/hevc/hm_common/c++/source_common
#include <cstring>
#include <exception>
#include <map>
#include <string>

// Entries come back as `width is_pair channel path\n` lines (maps iterate in key order); returns the number of entries,
// -1 if the file cannot be opened, -2 if the reference threw (std::stoul / vector::at on a malformed line).
extern "C" int ref_parse_three_keys(const char* path, const char* delimiters, char* out, int out_len)
{
    std::map<std::pair<unsigned int, unsigned int>, std::string> single, pair;
    try {
        if (parse_file_strings_three_keys(single, pair, path, delimiters) < 0) return -1;
    } catch (const std::exception&) {
        return -2;
    }
    std::string text;
    int n = 0;
    for (int p = 0; p < 2; p++)
        for (const auto& kv : (p ? pair : single)) {
            text += std::to_string(kv.first.first) + " " + std::to_string(p) + " " + std::to_string(kv.first.second) + " " + kv.second + "\n";
            n++;
        }
    if ((int)text.size() + 1 > out_len) return -3;
    memcpy(out, text.c_str(), text.size() + 1);
    return n;
}
