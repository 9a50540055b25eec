#include "lvm_internal.h"
namespace lvm {
int color_process(Ctx* c, const lvm_params&, int, const FrameIO&, hipStream_t, int* produced) { *produced = 0; c->err = "color: not built yet"; return LVM_ERR_INVALID; }
}
