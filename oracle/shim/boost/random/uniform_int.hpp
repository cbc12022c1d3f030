#pragma once
#include "boost/random.hpp"
