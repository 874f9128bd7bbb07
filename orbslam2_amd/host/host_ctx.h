// host_ctx.h -- the per-thread C-ABI context the host classes share (defined in ORBmatcher.cc).
#pragma once
#include "../../include/orbmatch.h"

namespace ORB_SLAM2 {
namespace detail {
orbm_ctx *HostCtx();                       // the calling thread's context; throws if the GPU context cannot be created
void HostCheck(int rc, const char *what);  // throws with the library's last error when rc < 0
orbm_ctx *NewCtx();                        // a context of its own for an object that is shared between threads (caller destroys it)
void CheckOn(orbm_ctx *ctx, int rc, const char *what);
}  // namespace detail
}  // namespace ORB_SLAM2
