// lasr_host.hip.h -- common prelude of the translation units of liblasr_hip.so: device code (lasr_kernels.hip.h: every kernel is a
// template or `inline`, so a unit emits only what it launches), the engine context and the launch interface.
#pragma once
#include "lasr_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/lasr.h"
#include "../../include/lasr_debug.h"

using namespace lasr;

#include "lasr_ctx.hip.h"
#include "lasr_launch.hip.h"
