// nrd_reblur_ortho.hip - the REBLUR / RELAX kernels compiled for orthographic projections (nrd_device.h NRD_ORTHO):
// same sources, nrdhip::ortho::launch_* entry points. The sample's "Ortho" camera: Source/NRDSample.cpp:1214, :1971.
#define NRD_ORTHO 1
#include "nrd_reblur.hip"
