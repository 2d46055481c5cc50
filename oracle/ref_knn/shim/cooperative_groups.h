#pragma once
#include <hip/hip_cooperative_groups.h>
