// dynamic_visual_slam/ORBextractor.hpp — source-compatible replacement for the reference header of the same include path
// (reference include/dynamic_visual_slam/ORBextractor.hpp:44-110): ORB_SLAM3::ORBextractor with the reference's
// constructor, operator(), scale getters and public mvImagePyramid, extracting on the MI355X through the C-ABI (dvs_orb_*).
// With this repo's include/ in front of the reference's, frontend.cpp:205-211, 1094-1095, 1285-1286 compile unchanged.
// (ExtractorNode, the quad-tree helper of the reference's .cpp, has no users outside ORBextractor.cpp and is not declared.)
#ifndef ORBEXTRACTOR_HPP
#define ORBEXTRACTOR_HPP
#ifndef DVSLAM_WITH_OPENCV
#define DVSLAM_WITH_OPENCV 1
#endif
#include "../dvslam/orb_extractor.hpp"
#endif  // ORBEXTRACTOR_HPP
