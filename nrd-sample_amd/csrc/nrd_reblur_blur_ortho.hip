// nrd_reblur_blur_ortho.hip - orthographic flavour of nrd_reblur_blur.hip (nrd_device.h NRD_ORTHO), nrdhip::ortho::launch_reblur_blur_radiance.
#define NRD_ORTHO 1
#define NRD_PART 1
#include "nrd_reblur.hip"
