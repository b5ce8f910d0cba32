// rank_scatter_small.hip -- 1-, 2- and 3-bit-digit instantiations: the rest of the reference's
// radix sweep (rs = {1,2,4,8}, .cu:1055-1062) and the multi-GPU MSB partition (top 1..3 bits).
#include "rank_scatter.hpp"

namespace lsd {

hipError_t launch_rank_scatter_small(int radix_bits, int rank_method, bool chained, const PassParams& p, hipStream_t stream)
{
    switch (radix_bits) {
        case 1: return launch_rank_scatter_shape<1, 256, 16>(rank_method, chained, p, stream);
        case 2: return launch_rank_scatter_shape<2, 256, 16>(rank_method, chained, p, stream);
        case 3: return launch_rank_scatter_shape<3, 256, 16>(rank_method, chained, p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lsd
