// bf16-storage 3x3 convolution, contraction width K = 128: the instances of conv3x3_bf16_impl.h's kernel template (one translation unit per K
// so that they build in parallel).  Design notes: conv3x3_bf16.hip.
#include "conv3x3_bf16_impl.h"

int ngan::conv3x3_bf16_launch_k128(ConvArgsB a, int N, int pgt, bool narrow, hipStream_t s) { return dispatch_n<128>(a, N, pgt, narrow, s); }
