// B2 on fp16 operand pairs ("f16x3" training): nf_mlp_bf16_dw.hip compiled with NFB_F16 = 1 -- see the note at its top.
#define NFB_F16 1
#include "nf_mlp_bf16_dw.hip"
