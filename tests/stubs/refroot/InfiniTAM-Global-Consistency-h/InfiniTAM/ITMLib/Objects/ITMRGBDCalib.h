// where the reference expects the InfiniTAM submodule's header: forwards to the mirror
#pragma once
#include "ITMLib/Objects/ITMRGBDCalib.h"
