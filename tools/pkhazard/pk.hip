#define STAGE_KERNEL stage_kernel_pk
#include "stage.h"
