// Library identification and error strings for include/qt_hip.h.
#include <string.h>
#include "qt_common.h"

extern "C" {

int qt_version(void) { return QT_VERSION_INT; }

const char* qt_target_arch(void) { return "gfx950"; }

const char* qt_strerror(int status) {
    switch (status) {
        case QT_OK: return "ok";
        case QT_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or inconsistent shape)";
        case QT_ERR_ALIGNMENT: return "alignment: packed planes need 16-byte aligned pointers and a row stride that is a multiple of 4 words (8 for nibble planes)";
        case QT_ERR_LAUNCH: return "HIP kernel launch failed";
        case QT_ERR_UNSUPPORTED: return "unsupported argument combination";
        case QT_ERR_NO_DEVICE: return "no gfx950 device available";
        default: return "unknown qt_status";
    }
}

int qt_device_info(char* name, int cap) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return QT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return QT_ERR_NO_DEVICE;
    if (name && cap > 0) {
        strncpy(name, prop.gcnArchName, (size_t)cap - 1);
        name[cap - 1] = '\0';
    }
    return prop.multiProcessorCount;
}

}  // extern "C"
