// One (epilogue, operand type) slice of conv3x3_bf16_kernel's instantiations -- see "translation units" in conv3x3_bf16.hip.
#define MAU_CONV_TU_EPI 2
#define MAU_CONV_TU_F16 1
#include "conv3x3_bf16.hip"
