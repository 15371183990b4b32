// Device policy and basic types of the C++ host side.  The reference selects its compute back end with
// a policy type (`struct Cpu` / `struct Gpu`, currennt_lib/src/Types.hpp:45-67) whose only content is
// the vector type; `Hip` is the third policy: its vectors live behind the C ABI of libcurrennt_hip.so
// (include/currennt_hip.h) and are mirrored to the host on demand.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/currennt_hip.h"

namespace currennt_hip {

typedef float real_t;                                  // Types.hpp:39

enum { PATTYPE_NONE = 0, PATTYPE_FIRST = 1, PATTYPE_NORMAL = 2, PATTYPE_LAST = 3 };   // Types.hpp:30-33

struct Hip {
    typedef std::vector<real_t> real_vector;           // host mirrors of device vectors
    typedef std::vector<int> int_vector;
    typedef std::vector<char> pattype_vector;
};

// turn a non-zero cn_status back into the exception the reference would have thrown
inline void hipCheck(int rc, cn_ctx *ctx = 0)
{
    if (rc != CN_OK) {
        const char *m = cn_last_error(ctx);
        throw std::runtime_error(m && *m ? std::string(m) : std::string("libcurrennt_hip error ") + std::to_string(rc));
    }
}

}  // namespace currennt_hip
