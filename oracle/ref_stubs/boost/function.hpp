// TEST INFRASTRUCTURE ONLY -- see ../README.md.  sdk/.../loop.h names boost::function<void ()>.
#pragma once
#include <functional>
namespace boost {
template <class Sig>
using function = std::function<Sig>;
}
