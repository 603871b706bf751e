#pragma once
#include "optix.h"
