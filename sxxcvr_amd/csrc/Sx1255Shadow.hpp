// SX1255 register shadow: the RF front-end control surface of the sx Device
// (tuning words, gain elements, antenna switches, raw register access) kept as
// values only.  In the reference these calls edit a register cache and push it
// to the chip over SPI (SoapySX.cpp:573-608, :1225-1561); here there is no
// chip, so the cache is the whole story, but it holds the same bit fields of
// the same registers, so applications and probing tools read back what the
// reference would have programmed.
//
// Table driven: every control is a (register, lowest bit, width) field.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace sx {

class Sx1255Shadow {
public:
    static constexpr unsigned kRegs = 0x80;

    struct Field {
        uint8_t addr, lsb, bits;
    };

    // One gain element: a field plus the dB range it spans (SoapySX.cpp:1291-1306).
    struct GainElement {
        const char *name;
        bool rx;
        Field field;
        double lo, hi, step;
        bool stepped_lna;       // the LNA field is not linear in dB (SoapySX.cpp:1320-1327, :1355)
    };

    explicit Sx1255Shadow(double master_clock) : clock_(master_clock)
    {
        // power-up image: registers 0x00-0x13 as the reference initialises them (datasheet defaults,
        // narrow RX filter, I2S dividers; SoapySX.cpp:145-176), then RX, TX and PA driver enabled (:625)
        static const uint8_t boot[] = {0x01, 0xD8, 0xF5, 0xC3, 0xD8, 0xF5, 0xC3, 0x11, 0x2E, 0x24,
                                       0x30, 0x02, 0x3F, 0x3B, 0x06, 0x00, 0x02, 0x00, 0x22, 0x2C};
        std::memset(regs_, 0, sizeof(regs_));
        std::memcpy(regs_, boot, sizeof(boot));
        put({0x00, 1, 3}, 7);
        tune(true, 433.92e6);
        tune(false, 433.92e6);
    }

    // ---- fields ---------------------------------------------------------------
    unsigned get(Field f) const
    {
        check(f.addr);
        return (regs_[f.addr] >> f.lsb) & ((1u << f.bits) - 1u);
    }

    void put(Field f, unsigned value)
    {
        check(f.addr);
        const unsigned mask = ((1u << f.bits) - 1u) << f.lsb;
        regs_[f.addr] = (uint8_t)((regs_[f.addr] & ~mask) | ((value << f.lsb) & mask));
    }

    // ---- synthesizers: 24-bit word in steps of master clock / 2^20 (SoapySX.cpp:1236-1272) -------
    void tune(bool rx, double hz)
    {
        const double step = clock_ / 1048576.0;
        const uint32_t word = (uint32_t)quantize(0.0, step * 16777215.0, step, hz);
        const uint8_t base = rx ? 0x01 : 0x04;
        put({base, 0, 8}, word >> 16);
        put({(uint8_t)(base + 1), 0, 8}, (word >> 8) & 0xFF);
        put({(uint8_t)(base + 2), 0, 8}, word & 0xFF);
    }

    double tuned(bool rx) const
    {
        const uint8_t base = rx ? 0x01 : 0x04;
        const uint32_t word = ((uint32_t)regs_[base] << 16) | ((uint32_t)regs_[base + 1] << 8) | regs_[base + 2];
        return (clock_ / 1048576.0) * word;
    }

    // ---- gains ----------------------------------------------------------------------------------
    static const std::vector<GainElement> &elements()
    {
        static const std::vector<GainElement> table = {
            {"LNA", true, {0x0C, 5, 3}, 0.0, 48.0, 6.0, true},
            {"PGA", true, {0x0C, 1, 4}, 0.0, 30.0, 2.0, false},
            {"DAC", false, {0x08, 4, 3}, 0.0, 9.0, 3.0, false},
            {"MIXER", false, {0x08, 0, 4}, 0.0, 30.0, 2.0, false},
        };
        return table;
    }

    static const GainElement *find(bool rx, const std::string &name)
    {
        for (const auto &e : elements())
            if (e.rx == rx && name == e.name) return &e;
        return nullptr;
    }

    void set_gain(bool rx, const std::string &name, double db)
    {
        const GainElement *e = find(rx, name);
        if (!e) return;                                       // unknown names are ignored, like the reference
        const int steps = quantize(e->lo, e->hi, e->step, db);
        put(e->field, e->stepped_lna ? lna_code(steps) : (unsigned)steps);
    }

    double gain(bool rx, const std::string &name) const
    {
        const GainElement *e = find(rx, name);
        if (!e) return 0.0;
        const unsigned code = get(e->field);
        const int steps = e->stepped_lna ? lna_steps(code) : (int)code;
        const double db = e->lo + e->step * steps;
        return db < e->lo ? e->lo : (db > e->hi ? e->hi : db);
    }

    // Overall gain: the coarse element is set first, aiming to leave `rest` dB for the fine one,
    // which then takes whatever is left (SoapySX.cpp:1370-1394: LNA/PGA with 12 dB, DAC/MIXER with 26 dB).
    void set_overall_gain(bool rx, double db)
    {
        const char *coarse = rx ? "LNA" : "DAC";
        const char *fine = rx ? "PGA" : "MIXER";
        const double rest = rx ? 12.0 : 26.0;
        set_gain(rx, coarse, db - rest);
        set_gain(rx, fine, db - gain(rx, coarse));
    }

    // ---- antenna switches (SoapySX.cpp:1416-1466): RX loop-back mode bits, TX PA driver enable -----
    void set_antenna(bool rx, const std::string &name)
    {
        if (rx) {
            if (name == "RX") put({0x10, 2, 2}, 0);
            else if (name == "LB") put({0x10, 2, 2}, 1);
            else if (name == "DLB") put({0x10, 2, 2}, 3);
        } else {
            if (name == "TX") put({0x00, 3, 1}, 1);
            else if (name == "NONE") put({0x00, 3, 1}, 0);
        }
    }

    std::string antenna(bool rx) const
    {
        if (!rx) return get({0x00, 3, 1}) ? "TX" : "NONE";
        const unsigned mode = get({0x10, 2, 2});
        return (mode & 2) ? "DLB" : ((mode & 1) ? "LB" : "RX");
    }

    // ---- raw access -------------------------------------------------------------------------------
    // A burst must stay inside the 7-bit register space (SoapySX.cpp:596-597).
    void write(unsigned addr, const std::vector<unsigned> &values)
    {
        for (size_t i = 0; i < values.size(); ++i) put({(uint8_t)std::min<size_t>(addr + i, 0xFF), 0, 8}, values[i]);
        if (addr >= kRegs || values.size() > kRegs || addr > kRegs - values.size())
            throw std::runtime_error("Invalid register address");
    }

    // Reads return the shadow; the status register reports both PLLs locked, which is what the
    // reference's start-up probe waits for (SoapySX.cpp:635-636).
    std::vector<unsigned> read(unsigned addr, size_t length) const
    {
        std::vector<unsigned> out(length, 0);
        for (size_t i = 0; i < length; ++i) {
            const unsigned r = (addr + i) & (kRegs - 1);
            out[i] = r == 0x11 ? 3u : regs_[r];
        }
        return out;
    }

private:
    static void check(unsigned addr)
    {
        if (addr >= kRegs) throw std::runtime_error("Invalid register address");
    }

    // clamp into [lo, hi], count steps from lo, round half away from zero (SoapySX.cpp:50-55)
    static int quantize(double lo, double hi, double step, double v)
    {
        v = v < lo ? lo : (v > hi ? hi : v);
        return (int)std::round((v - lo) / step);
    }

    // LNA: 0..6 steps -> code 6 - steps/2, 7 steps -> 2, 8 steps -> 1; read back through the inverse map
    static unsigned lna_code(int steps) { return steps <= 6 ? (unsigned)(6 - steps / 2) : (steps == 7 ? 2u : 1u); }
    static int lna_steps(unsigned code)
    {
        static const int inverse[8] = {0, 8, 7, 6, 4, 2, 0, 0};
        return inverse[code & 7];
    }

    double clock_;
    uint8_t regs_[kRegs];
};

}  // namespace sx
