// Host-only helpers shared by pnn_abi.cpp and pnn_host.cpp (no HIP): file reading and the model-table parser.
#pragma once
#include <string>
#include <vector>

namespace pnn {

bool read_file(const std::string& path, std::vector<char>* out);

struct TableEntry { int width, is_pair, channel; std::string path; };
// hevc/hm_common/c++/source_common/tools.cpp:52-111; returns PNN_OK or PNN_E_IO with *err set.
int parse_table(const char* path, std::vector<TableEntry>* out, std::string* err);
void set_create_error(const std::string& msg);      // what pnn_last_error(NULL) reports (defined in pnn_abi.cpp)

}  // namespace pnn
