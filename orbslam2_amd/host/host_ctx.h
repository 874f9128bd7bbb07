// host_ctx.h -- the per-thread C-ABI context the host classes share (defined in ORBmatcher.cc).
#pragma once
#include "../../include/orbmatch.h"

namespace ORB_SLAM2 {
namespace detail {
orbm_ctx *HostCtx();                       // throws if the GPU context cannot be created
void HostCheck(int rc, const char *what);  // throws with the library's last error when rc < 0
}  // namespace detail
}  // namespace ORB_SLAM2
