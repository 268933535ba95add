// nrd_sigma_ortho.hip - the SIGMA / REFERENCE kernels compiled for orthographic projections (nrd_device.h NRD_ORTHO):
// same sources, nrdhip::ortho::launch_* entry points.
#define NRD_ORTHO 1
#include "nrd_sigma.hip"
