// Error reporting shared by the translation units of libbn_hip.so.
#pragma once

#include <string>

namespace bn {

// Records msg as the calling thread's last error (what bn_last_error returns) and returns code.
int fail(int code, const std::string& msg);

}  // namespace bn
