// host-only sanitizer build: the two error helpers fs_capi.hip defines (that file needs a HIP device)
#include <string>
static thread_local std::string g_err;
void fs_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *fs_last_error(void) { return g_err.c_str(); }
extern "C" int fs_version(void) { return 100; }
