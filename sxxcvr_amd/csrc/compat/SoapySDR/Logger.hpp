// Minimal API-compatible subset of <SoapySDR/Logger.hpp> (see Constants.h).
#pragma once
#include <cstdarg>
#include <string>

#include "Constants.h"

typedef enum {
    SOAPY_SDR_FATAL = 1,
    SOAPY_SDR_CRITICAL = 2,
    SOAPY_SDR_ERROR = 3,
    SOAPY_SDR_WARNING = 4,
    SOAPY_SDR_NOTICE = 5,
    SOAPY_SDR_INFO = 6,
    SOAPY_SDR_DEBUG = 7,
    SOAPY_SDR_TRACE = 8,
    SOAPY_SDR_SSI = 9
} SoapySDRLogLevel;

typedef void (*SoapySDRLogHandler)(const SoapySDRLogLevel logLevel, const char *message);

extern "C" {
void SoapySDR_log(const SoapySDRLogLevel logLevel, const char *message);
void SoapySDR_vlogf(const SoapySDRLogLevel logLevel, const char *format, va_list argList);
void SoapySDR_logf(const SoapySDRLogLevel logLevel, const char *format, ...)
    __attribute__((format(printf, 2, 3)));
void SoapySDR_registerLogHandler(const SoapySDRLogHandler handler);
void SoapySDR_setLogLevel(const SoapySDRLogLevel logLevel);
SoapySDRLogLevel SoapySDR_getLogLevel(void);
}

namespace SoapySDR {
typedef SoapySDRLogLevel LogLevel;
typedef SoapySDRLogHandler LogHandler;
void log(const LogLevel logLevel, const std::string &message);
void vlogf(const SoapySDRLogLevel logLevel, const char *format, va_list argList);
void logf(const SoapySDRLogLevel logLevel, const char *format, ...) __attribute__((format(printf, 2, 3)));
void registerLogHandler(const LogHandler &handler);
void setLogLevel(const LogLevel logLevel);
}  // namespace SoapySDR
