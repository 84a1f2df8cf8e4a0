// TEST-ONLY (see ../core/core.hpp): cv::DMatch / cv::KeyPoint live in the core stand-in
#pragma once
#include "../core/core.hpp"
