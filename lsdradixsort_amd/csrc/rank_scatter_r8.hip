// rank_scatter_r8.hip -- 8-bit-digit instantiations of the rank-and-scatter kernel
// (BASELINE.json configs[2] and configs[4]).  Shape ids index kShapesR8 in aux_kernels.hip.
#include "rank_scatter.hpp"

namespace lsd {

hipError_t launch_rank_scatter_r8(int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream)
{
    switch (shape_id) {
        case 0: return launch_rank_scatter_shape<8, 512, 16>(rank_method, chained, p, stream);
        case 1: return launch_rank_scatter_shape<8, 256, 16>(rank_method, chained, p, stream);
        case 2: return launch_rank_scatter_shape<8, 1024, 16>(rank_method, chained, p, stream);
        case 3: return launch_rank_scatter_shape<8, 512, 32>(rank_method, chained, p, stream);
        case 4: return launch_rank_scatter_shape<8, 1024, 32>(rank_method, chained, p, stream);
        case 5: return launch_rank_scatter_shape<8, 512, 32, 8192>(rank_method, chained, p, stream);
        case 6: return launch_rank_scatter_shape<8, 512, 64, 8192>(rank_method, chained, p, stream);
        case 7: return launch_rank_scatter_shape<8, 512, 64, 16384>(rank_method, chained, p, stream);
        case 8: return launch_rank_scatter_shape<8, 1024, 32, 16384>(rank_method, chained, p, stream);
        case 9: return launch_rank_scatter_shape<8, 1024, 32, 8192>(rank_method, chained, p, stream);
        case 10: return launch_rank_scatter_shape<8, 256, 64, 8192>(rank_method, chained, p, stream);
        case 11: return launch_rank_scatter_shape<8, 512, 24, 4096>(rank_method, chained, p, stream);
        case 12: return launch_rank_scatter_shape<8, 512, 20, 2048>(rank_method, chained, p, stream);
        case 13: return launch_rank_scatter_shape<8, 1024, 16, 8192>(rank_method, chained, p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lsd
