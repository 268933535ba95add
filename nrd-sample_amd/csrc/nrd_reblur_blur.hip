// nrd_reblur_blur.hip - the non-SH REBLUR Blur kernels in their own translation unit (nrd_reblur.hip NRD_PART 1): same sources
// and flags, built side by side with the rest (make -j); entry point nrdhip::persp::launch_reblur_blur_radiance.
#define NRD_PART 1
#include "nrd_reblur.hip"
