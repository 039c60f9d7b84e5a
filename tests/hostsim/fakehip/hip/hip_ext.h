// TEST INFRASTRUCTURE ONLY: see hip_runtime.h beside this file (hipExtLaunchKernelGGL is defined there).
#pragma once
#include "hip_runtime.h"
