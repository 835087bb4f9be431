// Error plumbing of libbxmi (core.hip) for the host-only sanitizer build of csrc/bedparse.cpp -- test infrastructure.
#include <cstdarg>
#include <cstdio>
#include <string>
namespace bxmi {
std::string &last_error()
{
    static thread_local std::string e;
    return e;
}
int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}
}  // namespace bxmi
