// rank_scatter_r8.hip -- 8-bit-digit instantiations of the rank-and-scatter kernel
// (BASELINE.json configs[2] and configs[4]).  Shape ids index kShapesR8 in aux_kernels.hip.
#include "rank_scatter.hpp"

namespace lsd {

hipError_t launch_rank_scatter_r8(int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream)
{
    switch (shape_id) {
        case 0: return launch_rank_scatter_shape<8, 512, 32, 16384>(rank_method, chained, p, stream);   // keys default
        case 1: return launch_rank_scatter_shape<8, 1024, 16, 8192>(rank_method, chained, p, stream);
        case 2: return launch_rank_scatter_shape<8, 1024, 32, 16384>(rank_method, chained, p, stream);
        case 3: return launch_rank_scatter_shape<8, 512, 16, 8192>(rank_method, chained, p, stream);
        case 4: return launch_rank_scatter_shape<8, 1024, 32, 32768>(rank_method, chained, p, stream);  // key/value default
        case 5: return launch_rank_scatter_shape<8, 256, 16, 4096>(rank_method, chained, p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lsd
