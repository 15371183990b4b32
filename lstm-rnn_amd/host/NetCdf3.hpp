// NetCDF-3 "classic" (CDF-1 / CDF-2) reader for CURRENNT data files (format: reference README:600-646;
// the reference links libnetcdf, data_sets/DataSet.cpp:44-144, which this image does not have).
// Reads the header and serves hyperslabs of fixed-size variables; big-endian on disk.
#pragma once

#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <stdint.h>

namespace currennt_hip {

class NetCdf3File {
public:
    enum NcType { NC_BYTE = 1, NC_CHAR = 2, NC_SHORT = 3, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6 };
    struct Var { std::string name; std::vector<int> dimids; int type; uint64_t vsize, begin; bool record; };

    explicit NetCdf3File(const std::string &filename) : m_f(fopen(filename.c_str(), "rb")), m_name(filename)
    {
        if (!m_f) throw std::runtime_error("Could not open '" + filename + "': No such file or directory");   // DataSet.cpp:482-483
        try { readHeader(); }
        catch (...) { fclose(m_f); throw; }
    }
    ~NetCdf3File() { if (m_f) fclose(m_f); }

    bool hasDimension(const std::string &n) const { return m_dimIndex.count(n) != 0; }
    int dimension(const std::string &n) const
    {
        std::map<std::string, int>::const_iterator it = m_dimIndex.find(n);
        if (it == m_dimIndex.end()) throw std::runtime_error("Cannot get dimension '" + n + "': NetCDF: Invalid dimension ID or name");
        return (int)m_dimLen[it->second];
    }
    bool hasVariable(const std::string &n) const { return m_varIndex.count(n) != 0; }

    // `count` elements starting at flat element index `start` of a fixed-size variable
    std::vector<float> readFloats(const std::string &var, uint64_t start, uint64_t count) const
    {
        const Var &v = variable(var);
        std::vector<float> out(count);
        if (v.type == NC_FLOAT) { std::vector<unsigned char> raw = readRaw(v, start, count, 4); for (uint64_t i = 0; i < count; ++i) out[i] = beFloat(&raw[4 * i]); }
        else if (v.type == NC_DOUBLE) { std::vector<unsigned char> raw = readRaw(v, start, count, 8); for (uint64_t i = 0; i < count; ++i) out[i] = (float)beDouble(&raw[8 * i]); }
        else throw std::runtime_error("Cannot read array '" + var + "': NetCDF: Not a valid data type");
        return out;
    }
    std::vector<int> readInts(const std::string &var, uint64_t start, uint64_t count) const
    {
        const Var &v = variable(var);
        if (v.type != NC_INT) throw std::runtime_error("Cannot read array '" + var + "': NetCDF: Not a valid data type");
        std::vector<unsigned char> raw = readRaw(v, start, count, 4);
        std::vector<int> out(count);
        for (uint64_t i = 0; i < count; ++i) out[i] = (int)be32(&raw[4 * i]);
        return out;
    }
    std::string readString(const std::string &var, uint64_t row, uint64_t maxLen) const
    {
        const Var &v = variable(var);
        if (v.type != NC_CHAR) throw std::runtime_error("Cannot read variable '" + var + "': NetCDF: Not a valid data type");
        std::vector<unsigned char> raw = readRaw(v, row * maxLen, maxLen, 1);
        std::string s((const char *)raw.data(), raw.size());
        return std::string(s.c_str());     // cut at the first NUL like the reference (DataSet.cpp:78-79)
    }

private:
    FILE *m_f;
    std::string m_name;
    bool m_64 = false;
    uint32_t m_numrecs = 0;
    std::vector<uint64_t> m_dimLen;
    std::map<std::string, int> m_dimIndex, m_varIndex;
    std::vector<Var> m_vars;

    static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
    static float beFloat(const unsigned char *p) { uint32_t u = be32(p); float f; memcpy(&f, &u, 4); return f; }
    static double beDouble(const unsigned char *p) { uint64_t u = ((uint64_t)be32(p) << 32) | be32(p + 4); double d; memcpy(&d, &u, 8); return d; }

    void need(size_t n, unsigned char *buf) const
    {
        if (fread(buf, 1, n, m_f) != n) throw std::runtime_error("'" + m_name + "': NetCDF: truncated file");
    }
    uint32_t rd32() const { unsigned char b[4]; need(4, b); return be32(b); }
    uint64_t rdOff() const { if (!m_64) return rd32(); uint64_t hi = rd32(); return (hi << 32) | rd32(); }
    std::string rdName() const
    {
        uint32_t n = rd32();
        std::vector<unsigned char> b((n + 3) / 4 * 4 + 1);
        if (n) need((n + 3) / 4 * 4, b.data());
        return std::string((const char *)b.data(), n);
    }
    static size_t typeSize(int t) { switch (t) { case NC_BYTE: case NC_CHAR: return 1; case NC_SHORT: return 2; case NC_INT: case NC_FLOAT: return 4; case NC_DOUBLE: return 8; } return 0; }
    void skipAttrs() const
    {
        uint32_t tag = rd32(), n = rd32();
        if (tag == 0 && n == 0) return;
        if (tag != 0x0C) throw std::runtime_error("'" + m_name + "': NetCDF: bad attribute list");
        for (uint32_t i = 0; i < n; ++i) {
            rdName();
            uint32_t type = rd32(), nelems = rd32();
            size_t bytes = (typeSize((int)type) * nelems + 3) / 4 * 4;
            fseek(m_f, (long)bytes, SEEK_CUR);
        }
    }
    void readHeader()
    {
        unsigned char magic[4]; need(4, magic);
        if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 1 && magic[3] != 2))
            throw std::runtime_error("Could not open '" + m_name + "': NetCDF: Unknown file format");
        m_64 = magic[3] == 2;
        m_numrecs = rd32();
        uint32_t tag = rd32(), n = rd32();
        if (!(tag == 0 && n == 0)) {
            if (tag != 0x0A) throw std::runtime_error("'" + m_name + "': NetCDF: bad dimension list");
            for (uint32_t i = 0; i < n; ++i) { std::string nm = rdName(); m_dimIndex[nm] = (int)m_dimLen.size(); m_dimLen.push_back(rd32()); }
        }
        skipAttrs();
        tag = rd32(); n = rd32();
        if (!(tag == 0 && n == 0)) {
            if (tag != 0x0B) throw std::runtime_error("'" + m_name + "': NetCDF: bad variable list");
            for (uint32_t i = 0; i < n; ++i) {
                Var v; v.name = rdName();
                uint32_t nd = rd32();
                v.record = false;
                for (uint32_t k = 0; k < nd; ++k) { int id = (int)rd32(); v.dimids.push_back(id); if (k == 0 && m_dimLen.at(id) == 0) v.record = true; }
                skipAttrs();
                v.type = (int)rd32(); v.vsize = rd32(); v.begin = rdOff();
                m_varIndex[v.name] = (int)m_vars.size(); m_vars.push_back(v);
            }
        }
    }
    const Var &variable(const std::string &n) const
    {
        std::map<std::string, int>::const_iterator it = m_varIndex.find(n);
        if (it == m_varIndex.end()) throw std::runtime_error("Cannot read array '" + n + "': NetCDF: Variable not found");
        if (m_vars[it->second].record) throw std::runtime_error("Cannot read array '" + n + "': record variables are not supported");
        return m_vars[it->second];
    }
    std::vector<unsigned char> readRaw(const Var &v, uint64_t start, uint64_t count, size_t esz) const
    {
        uint64_t total = 1;
        for (size_t k = 0; k < v.dimids.size(); ++k) total *= m_dimLen[v.dimids[k]];
        if (start + count > total) throw std::runtime_error("Cannot read array '" + v.name + "': NetCDF: Index exceeds dimension bound");
        std::vector<unsigned char> raw(count * esz + 1);
        if (fseeko(m_f, (off_t)(v.begin + start * esz), SEEK_SET) != 0 || fread(raw.data(), 1, count * esz, m_f) != count * esz)
            throw std::runtime_error("Cannot read array '" + v.name + "': NetCDF: read error");
        raw.resize(count * esz);
        return raw;
    }
};

}  // namespace currennt_hip
