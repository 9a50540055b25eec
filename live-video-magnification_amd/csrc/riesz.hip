#include "lvm_internal.h"
namespace lvm {
int riesz_process(Ctx* c, const lvm_params&, int, const FrameIO&, hipStream_t, int* produced) { *produced = 0; c->err = "riesz: not built yet"; return LVM_ERR_INVALID; }
}
