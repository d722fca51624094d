#include "ros.h"
