#define STAGE_KERNEL stage_kernel_nopk
#include "stage.h"
