// Minimal API-compatible subset of <SoapySDR/Registry.hpp> (see Constants.h).
#pragma once
#include <map>
#include <string>

#include "Types.hpp"

namespace SoapySDR {

class Device;

typedef KwargsList (*FindFunction)(const Kwargs &);
typedef Device *(*MakeFunction)(const Kwargs &);
typedef std::map<std::string, FindFunction> FindFunctions;
typedef std::map<std::string, MakeFunction> MakeFunctions;

class Registry {
public:
    Registry(const std::string &name, const FindFunction &find, const MakeFunction &make, const std::string &abi);
    ~Registry(void);
    static FindFunctions listFindFunctions(void);
    static MakeFunctions listMakeFunctions(void);

private:
    std::string _name;
};

}  // namespace SoapySDR
