// rank_scatter_r4.hip -- 4-bit-digit instantiations of the rank-and-scatter kernel
// (BASELINE.json configs[1]).  Shape ids index kShapesR4 in aux_kernels.hip.
#include "rank_scatter.hpp"

namespace lsd {

hipError_t launch_rank_scatter_r4(int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream)
{
    switch (shape_id) {
        case 0: return launch_rank_scatter_shape<4, 512, 32, 16384>(rank_method, chained, p, stream);
        case 1: return launch_rank_scatter_shape<4, 512, 16, 8192>(rank_method, chained, p, stream);
        case 2: return launch_rank_scatter_shape<4, 256, 16, 4096>(rank_method, chained, p, stream);
        case 3: return launch_rank_scatter_shape<4, 1024, 32, 16384>(rank_method, chained, p, stream);
        case 4: return launch_rank_scatter_shape<4, 1024, 32, 32768>(rank_method, chained, p, stream);
        case 5: return launch_rank_scatter_shape<4, 1024, 16, 16384>(rank_method, chained, p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lsd
