// sw_internal.hpp — what the translation units of libcudasw4_amd.so share beside the public header
#pragma once
#include <cstdint>
#include <string>

struct sw_ctx;

namespace swi {
// record the thread's error message (sw_last_error) and return `code`
int fail(int code, const std::string& msg);
int32_t query_length(const sw_ctx* ctx);   // 0: no query installed
int device_of(const sw_ctx* ctx);
int num_cus(const sw_ctx* ctx);
}  // namespace swi
