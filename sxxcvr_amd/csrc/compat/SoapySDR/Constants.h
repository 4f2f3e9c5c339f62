// Minimal API-compatible subset of <SoapySDR/Constants.h> / <SoapySDR/Errors.h>.
// Used ONLY when the real SoapySDR development headers are absent (they are
// absent in this image); with SoapySDR installed, drop compat/ from the
// include path and the plugin compiles against the real headers unchanged.
// Values follow the public SoapySDR 0.8 API.
#pragma once

#define SOAPY_SDR_TX 0
#define SOAPY_SDR_RX 1

#define SOAPY_SDR_END_BURST (1 << 1)
#define SOAPY_SDR_HAS_TIME (1 << 2)
#define SOAPY_SDR_END_ABRUPT (1 << 3)
#define SOAPY_SDR_ONE_PACKET (1 << 4)
#define SOAPY_SDR_MORE_FRAGMENTS (1 << 5)
#define SOAPY_SDR_WAIT_TRIGGER (1 << 6)

#define SOAPY_SDR_TIMEOUT (-1)
#define SOAPY_SDR_STREAM_ERROR (-2)
#define SOAPY_SDR_CORRUPTION (-3)
#define SOAPY_SDR_OVERFLOW (-4)
#define SOAPY_SDR_NOT_SUPPORTED (-5)
#define SOAPY_SDR_TIME_ERROR (-6)
#define SOAPY_SDR_UNDERFLOW (-7)

#define SOAPY_SDR_CF32 "CF32"
#define SOAPY_SDR_CS16 "CS16"
#define SOAPY_SDR_CF16 "CF16"

#define SOAPY_SDR_ABI_VERSION "0.8-compat"
#define SOAPY_SDR_API
