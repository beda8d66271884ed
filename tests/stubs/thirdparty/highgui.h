// stand-in: <highgui.h> (OpenCV 1 style include of the reference's Input.h); the types live in opencv/cv.h
#pragma once
#include <opencv/cv.h>
