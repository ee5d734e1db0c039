// the double / KS_MULLER contexts (see nrs_ctx_impl.h)
#include "nrs_ctx_impl.h"
namespace nrs {
template CtxBase *make_ctx2<double, KS_MULLER>(bool);
}
