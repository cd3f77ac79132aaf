// test infrastructure: see ../../opencv/cv.h
#pragma once
#include <opencv/cv.h>
