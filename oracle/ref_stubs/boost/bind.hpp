// TEST INFRASTRUCTURE ONLY -- see ../README.md.  Included by unitree_legged_sdk.h, never used on our path.
#pragma once
#include <functional>
