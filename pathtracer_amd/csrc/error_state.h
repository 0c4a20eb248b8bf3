// Thread-local error text behind ptamd_last_error().
#pragma once
#include <string>

namespace ptamd {
void set_error(const std::string &msg);
const char *get_error();
}  // namespace ptamd
