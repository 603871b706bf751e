// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// L2 — placeFeature / placeCaveFeature (featurePlacement.hpp:147-1379).  PLACEHOLDER: filled in by the feature milestone.
#include "mmo_stages.h"
namespace mmo {
bool placeFeature(const FeaturePlacement&, ivec3, Block*) { return false; }
bool placeCaveFeature(const CaveFeaturePlacement&, ivec3, Block*) { return false; }
}
