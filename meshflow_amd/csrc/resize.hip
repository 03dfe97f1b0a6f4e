// Crop + bilinear resize (reference: meshflowstabilizer.py:1111-1157, cv2.resize INTER_LINEAR).
// Placeholder until the resize row is implemented: fails loudly.
#include "mf_common.h"

namespace mf {

int launch_crop_resize(const uint8_t*, uint8_t*, int, int, int, int, int, int, int, hipStream_t)
{
    set_error("mf_crop_resize_u8c3: not implemented yet");
    return MF_ERR_INVALID_ARG;
}

}  // namespace mf
