// maintain.hip -- voxel decay, sliding window and swapping (placeholder while the core path is brought up).
#include "dslam_internal.h"

namespace dslam {
static int unsupported(const char *what) { set_last_error(std::string(what) + " not implemented yet"); return DSLAM_ERR_UNSUPPORTED; }
int launch_decay(dslam_engine *, dslam_scene *, dslam_render_state *, int, int, int, int) { return unsupported("decay"); }
int launch_slide_pop(dslam_engine *, dslam_scene *, dslam_render_state *, int) { return unsupported("slide window"); }
int launch_swap_in(dslam_engine *, dslam_scene *, dslam_render_state *) { return unsupported("swap in"); }
int launch_swap_out(dslam_engine *, dslam_scene *, dslam_render_state *, bool) { return unsupported("swap out"); }
int launch_save_to_global(dslam_engine *, dslam_scene *) { return unsupported("save to global"); }
}  // namespace dslam
