// simulation.hpp -- the reference's type-erased `decoder` (src/simulation/simulation.h:23-69) on top of the
// facade, so that code written against `class decoder` -- awgn_simulation / bitflip_simulation
// (simulation.c++:95-213), the registry of benchmark.c++:23-166 -- takes a GPU-backed code object unchanged:
//
//     decoder d(cyclic::primitive_bch<5, dmin<7>, cyclic::berlekamp_massey_tag>());
//     std::vector<float> b(d.n());  ...                      // simulation.c++:100,:125
//     auto result = d.correct(b);                            // std::vector<math::ef_element<2, 1>>, or throws
//     d.to_string(); d.rate(); d.n();
//
// Same four virtuals (correct, to_string, rate, n), same value semantics (copies share the immutable code
// object), same exceptions (decoding_failure, std::runtime_error).  One call is one kernel launch and two small
// copies: see INTEGRATION.md section 1 for what that costs per frame and `correct_batch` below for the form
// that decodes a whole Eb/N0 point at once (the frames of simulation.c++:124-136 are independent).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "channelcoding_amd/cyclic.hpp"

class decoder {
 public:
  using return_type = math::ef_element<2, 1>;

 private:
  class decoder_concept {
   public:
    virtual ~decoder_concept() = default;
    virtual std::vector<return_type> correct(const std::vector<float> &b) const = 0;
    virtual std::string to_string() const = 0;
    virtual double rate() const = 0;
    virtual unsigned n() const = 0;
    // not in the reference: B frames of n soft values -> B * n hard decisions and one CC_FRAME_* status per frame
    virtual void correct_batch(const float *y, size_t B, std::vector<uint8_t> &words, std::vector<int32_t> &status) const = 0;
  };
  template <typename T> class decoder_model : public decoder_concept {
    T implementation;

   public:
    explicit decoder_model(T arg) : implementation(std::move(arg)) {}
    std::vector<return_type> correct(const std::vector<float> &b) const override {
      return implementation.template correct<return_type>(b);
    }
    std::string to_string() const override { return implementation.to_string(); }
    double rate() const override { return implementation.rate; }
    unsigned n() const override { return implementation.n; }
    void correct_batch(const float *y, size_t B, std::vector<uint8_t> &words, std::vector<int32_t> &status) const override {
      auto r = implementation.correct_batch(y, B);
      words = std::move(r.words);
      status = std::move(r.status);
    }
  };

  std::shared_ptr<const decoder_concept> _self;

 public:
  template <typename T> decoder(T decoder_) : _self(std::make_shared<decoder_model<T>>(std::move(decoder_))) {}

  template <typename InputSequence> std::vector<return_type> correct(const InputSequence &b) const {
    return _self->correct(b);
  }
  std::string to_string() const { return _self->to_string(); }
  double rate() const { return _self->rate(); }
  unsigned n() const { return _self->n(); }
  void correct_batch(const float *y, size_t B, std::vector<uint8_t> &words, std::vector<int32_t> &status) const {
    _self->correct_batch(y, B, words, status);
  }
};
