// sahs_model.hpp -- which field architecture a translation unit is built for.
//
// The field sources (sahs_layout.hpp, pack.hip, field_f32.hip) are written once against the architecture constants below
// and compiled once per model (build.py passes -DSAHS_MODEL=n); each build lives in its own namespace and exports its
// launchers under its own suffix, so both sit in one libsahs_nerf.so.
//   0  AudioFaceModel, config/audio/*.yml            (models.py:381-528)   namespace sahs,    launchers  name
//   1  NeRFaceModel, config/expression/person_2|3.yml (models.py:189-378)   namespace sahs_nf, launchers  name_nf
//   2  NeRFaceModel without deformation, config/expression/person_1.yml (use_warp False, use_ambient False)
//                                                                          namespace sahs_ns, launchers  name_ns
#pragma once
#ifndef SAHS_MODEL
#define SAHS_MODEL 0
#endif
#if SAHS_MODEL == 0
#define SAHS_NS sahs
#define SAHS_SYM(name) name
#elif SAHS_MODEL == 1
#define SAHS_NS sahs_nf
#define SAHS_SYM(name) name##_nf
#elif SAHS_MODEL == 2
#define SAHS_NS sahs_ns
#define SAHS_SYM(name) name##_ns
#else
#error "unknown SAHS_MODEL"
#endif
