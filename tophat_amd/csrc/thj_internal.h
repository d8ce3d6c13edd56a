// thj_internal.h -- declarations shared by the translation units of libthj_hip.so
#pragma once
void thj_set_error(const char* fmt, ...);
