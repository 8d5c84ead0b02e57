// B1 of the second model family on fp16 operand pairs: nf_mlp_lcode_bf16_bwd.hip compiled with NFB_F16 = 1.
#define NFB_F16 1
#include "nf_mlp_lcode_bf16_bwd.hip"
