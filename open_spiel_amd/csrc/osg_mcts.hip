// MCTS on the device (placeholder until the wavefront search lands this round).
#include "osg_internal.h"
extern "C" int osg_mcts_search(const osg_batch*, const osg_mcts_cfg*, int32_t*, int32_t*, double*, int8_t*, double*, int) {
  return osg::set_error(OSG_ERR_UNSUPPORTED, "osg_mcts_search: not implemented yet");
}
