// rank_scatter_small.hip -- 1-, 2- and 3-bit-digit instantiations: the rest of the reference's
// radix sweep (rs = {1,2,4,8}, .cu:1055-1062) and the multi-GPU MSB partition (top 1..3 bits).
#include "rank_scatter.hpp"

namespace lsd {

template <int R>
static hipError_t launch_small(int shape_id, int rank_method, bool chained, const PassParams& p, hipStream_t stream)
{
    switch (shape_id) {
        case 0: return launch_rank_scatter_shape<R, 256, 16>(rank_method, chained, p, stream);
        case 1: return launch_rank_scatter_shape<R, 512, 32, 16384>(rank_method, chained, p, stream);
        case 2: return launch_rank_scatter_shape<R, 1024, 32, 32768>(rank_method, chained, p, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_rank_scatter_small(int radix_bits, int shape_id, int rank_method, bool chained, const PassParams& p,
                                     hipStream_t stream)
{
    switch (radix_bits) {
        case 1: return launch_small<1>(shape_id, rank_method, chained, p, stream);
        case 2: return launch_small<2>(shape_id, rank_method, chained, p, stream);
        case 3: return launch_small<3>(shape_id, rank_method, chained, p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lsd
