// Tabular CFR on the device (placeholder until the kernels land this round).
#include "osg_internal.h"
#define NYI(name) return osg::set_error(OSG_ERR_UNSUPPORTED, name ": not implemented yet")
extern "C" {
int osg_cfr_create(osg_ctx*, const char*, const osg_cfr_cfg*, osg_cfr**) { NYI("osg_cfr_create"); }
int osg_cfr_destroy(osg_cfr*) { return OSG_OK; }
int osg_cfr_sizes(const osg_cfr*, int64_t*) { NYI("osg_cfr_sizes"); }
int osg_cfr_iterate(osg_cfr*, int) { NYI("osg_cfr_iterate"); }
int osg_mccfr_iterate(osg_cfr*, uint64_t, int64_t, int64_t) { NYI("osg_mccfr_iterate"); }
int osg_cfr_table_ptrs(osg_cfr*, double**, double**, double**) { NYI("osg_cfr_table_ptrs"); }
int osg_mccfr_delta_ptrs(osg_cfr*, double**, double**) { NYI("osg_mccfr_delta_ptrs"); }
int osg_mccfr_apply_deltas(osg_cfr*) { NYI("osg_mccfr_apply_deltas"); }
int osg_cfr_tables(const osg_cfr*, int32_t*, int32_t*, double*, double*, double*, double*) { NYI("osg_cfr_tables"); }
int osg_cfr_infostate_key(const osg_cfr*, int64_t, char*, int) { NYI("osg_cfr_infostate_key"); }
}
