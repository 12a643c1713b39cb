// smplr_build_id(): the sha256 the Makefile took over csrc/*.hip, csrc/*.h and include/smplraster.h when this
// library was built (host-only translation unit; the id arrives as -DSMPLR_BUILD_ID).
#ifndef SMPLR_BUILD_ID
#error "build through csrc/Makefile: it passes -DSMPLR_BUILD_ID"
#endif
extern "C" const char *smplr_build_id(void) { return SMPLR_BUILD_ID; }
