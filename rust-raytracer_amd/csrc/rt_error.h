// rt_error.h -- per-thread last-error message behind rt_last_error().
#pragma once

#include <string>

namespace rt {

inline std::string &last_error_slot() {
    static thread_local std::string msg;
    return msg;
}

// Records `msg` and returns `code`, so call sites read `return set_error(...)`.
inline int set_error(int code, const std::string &msg) {
    last_error_slot() = msg;
    return code;
}

}  // namespace rt
