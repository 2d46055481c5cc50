#pragma once
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
