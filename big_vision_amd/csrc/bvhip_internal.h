// Internal: pulls in the public C ABI so kernels and the header cannot drift.
#pragma once
#include "../../include/bvhip.h"
