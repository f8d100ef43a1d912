// The C ABI's error transport, shared by every translation unit that defines entry points: gtx::Error -> status code +
// thread-local message (gtx_last_error()). No C++ exception crosses the boundary.
#pragma once
#include <string>

#include "../../include/gtx.h"
#include "common.hpp"

namespace gtx {
inline thread_local std::string g_last_error;

template <typename F>
int guarded(F&& f) {
  try {
    f();
    g_last_error.clear();
    return GTX_OK;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return GTX_ERR_INTERNAL;
  } catch (...) {
    g_last_error = "unknown exception";
    return GTX_ERR_INTERNAL;
  }
}
}  // namespace gtx
