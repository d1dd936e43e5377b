// gen_stdlib_vectors.cpp -- golden vectors for the libstdc++ <random> distributions the
// reference calls (src/util/random.hpp:68-78).  libstdc++ is a third-party dependency that is
// not part of /root/reference but IS in this image (GCC 11.4), so the vectors come from the real
// implementation: a minimal PCG32 URBG (written here from the PCG paper's XSH-RR definition, with
// the reference's output-then-advance order) drives std::uniform_real_distribution<float>,
// <double> and std::uniform_int_distribution<size_t>.
//
//   g++ -std=c++17 -O2 tests/golden/gen_stdlib_vectors.cpp -o /tmp/gen_stdlib && /tmp/gen_stdlib > tests/golden/stdlib_vectors.json
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <random>

struct Pcg32 {
	typedef uint32_t result_type;
	uint64_t state, inc;
	static constexpr result_type min() { return std::numeric_limits<result_type>::min(); }
	static constexpr result_type max() { return std::numeric_limits<result_type>::max(); }
	result_type operator()() {
		uint32_t xs = (uint32_t)(((state >> 18u) ^ state) >> 27u);
		uint32_t rot = (uint32_t)(state >> 59u);
		uint32_t r = (xs >> rot) | (xs << ((0u - rot) & 31u));
		state = state * 6364136223846793005ull + inc;
		return r;
	}
};

int main() {
	const uint64_t seeds[][2] = { { 0x9DCE13F59DCE13F5ull, 0x9DCE13F59DCE13F5ull }, { 0x853C49E6748FEA9Bull, 0xDA3E39CB94B95BDBull },
	                              { 1ull, 1ull }, { 0xFFFFFFFFFFFFFFFFull, 0x1234567ull } };
	printf("{\n \"cases\": [\n");
	for (size_t s = 0; s < 4; ++s) {
		printf("  {\"state\": \"%" PRIu64 "\", \"inc\": \"%" PRIu64 "\",\n", seeds[s][0], seeds[s][1]);
		{
			Pcg32 g{ seeds[s][0], seeds[s][1] };
			printf("   \"u32\": [");
			for (int i = 0; i < 8; ++i) printf("%s%u", i ? ", " : "", g());
			printf("],\n");
		}
		{
			Pcg32 g{ seeds[s][0], seeds[s][1] };
			printf("   \"rand_1f_hex\": [");
			for (int i = 0; i < 16; ++i) { float f = std::uniform_real_distribution<float>()(g); printf("%s\"%a\"", i ? ", " : "", (double)f); }
			printf("],\n");
		}
		{
			Pcg32 g{ seeds[s][0], seeds[s][1] };
			printf("   \"rand_1d_hex\": [");
			for (int i = 0; i < 16; ++i) { double f = std::uniform_real_distribution<double>()(g); printf("%s\"%a\"", i ? ", " : "", f); }
			printf("],\n");
		}
		const size_t ranges[] = { 1, 2, 3, 6, 7, 1000, 4000000000ull };
		printf("   \"rand_choice\": {");
		for (size_t ri = 0; ri < sizeof ranges / sizeof ranges[0]; ++ri) {
			Pcg32 g{ seeds[s][0], seeds[s][1] };
			printf("%s\"%zu\": [", ri ? ", " : "", ranges[ri]);
			for (int i = 0; i < 24; ++i) {
				std::uniform_int_distribution<size_t> d(0, ranges[ri] - 1);
				printf("%s%zu", i ? ", " : "", d(g));
			}
			printf("]");
		}
		printf("},\n");
		{
			// draws consumed: state after 24 choices with n=1 and n=6
			Pcg32 g{ seeds[s][0], seeds[s][1] };
			for (int i = 0; i < 24; ++i) { std::uniform_int_distribution<size_t> d(0, 0); d(g); }
			printf("   \"state_after_24_choice1\": \"%" PRIu64 "\",\n", g.state);
			Pcg32 h{ seeds[s][0], seeds[s][1] };
			for (int i = 0; i < 24; ++i) { std::uniform_int_distribution<size_t> d(0, 5); d(h); }
			printf("   \"state_after_24_choice6\": \"%" PRIu64 "\"\n", h.state);
		}
		printf("  }%s\n", s + 1 < 4 ? "," : "");
	}
	printf(" ]\n}\n");
	return 0;
}
