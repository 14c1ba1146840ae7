// galois.cpp -- explicit instantiations of the templates in galois.hpp for the two symbol widths.
#include "galois.hpp"

namespace ccamd {

template struct FieldT<uint8_t>;
template struct FieldT<uint16_t>;
template CodeTablesT<uint8_t> build_code<uint8_t>(const FieldT<uint8_t> &, int, unsigned, unsigned, unsigned);
template CodeTablesT<uint16_t> build_code<uint16_t>(const FieldT<uint16_t> &, int, unsigned, unsigned, unsigned);

}  // namespace ccamd
