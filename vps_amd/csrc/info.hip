// Library identification for the ctypes loader.
#include "common.h"

#define VPS_ABI_VERSION 18

extern "C" int vps_abi_version(void) { return VPS_ABI_VERSION; }

extern "C" const char* vps_build_info(void) {
#define VPS_STR2(x) #x
#define VPS_STR(x) VPS_STR2(x)
    return "libvpship abi=" VPS_STR(VPS_ABI_VERSION) " arch=gfx950 wave=64 mfma=f32_32x32x2,bf16_32x32x16(split x3/x6),f16_32x32x16(split x3) "
           "kernels=conv_mfma,flow_ops,nn_ops,det_ops,head_ops,pan_ops,post_ops";
}
