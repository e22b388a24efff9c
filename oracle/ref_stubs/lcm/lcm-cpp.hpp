// TEST INFRASTRUCTURE ONLY -- see ../README.md.  sdk/.../lcm.h declares members of these types and reads rbuf->data.
#pragma once
#include <string>
namespace lcm {
struct ReceiveBuffer {
  void *data;
  unsigned int data_size;
  long long recv_utime;
};
class Subscription;
class LCM {};
}  // namespace lcm
