// tests/refdrop: force-included in front of the reference's terrain.cpp ONLY.  terrain.cpp:437 throws `std::exception("invalid offset")`,
// a constructor that exists in MSVC's STL only.  Every standard / glm header the translation unit uses is included here first, then the
// token `exception` is pointed at std::runtime_error for the rest of the file (its one other use would be none).
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <exception>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <hip/hip_runtime_api.h>
#include <glm/glm.hpp>
#include <glm/gtx/string_cast.hpp>
#define exception runtime_error
