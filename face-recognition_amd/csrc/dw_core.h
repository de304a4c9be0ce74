// Shared depthwise types.
#pragma once
#include "kernels.h"

template <typename T>
struct V8 { T v[8]; };   // 8 channels as stored (16 B for bf16, 32 B for fp32)
