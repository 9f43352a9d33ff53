// traps.h -- the header a maintainer of rampa069/PhnRec puts in place of the reference's traps.h
// (option (a)): the PUBLIC interface of class Traps as SpeechRec uses it
// (reference traps.h:59-75: Init / CalcFeatures / CalcFeaturesBunched / Reset / Get* / Set*), with the
// scratch buffers and the three NeuralNet members replaced by one opaque handle of the MI355X library.
// Same include guard as the reference's header, so that `-include traps.h` in front of the reference's
// translation units (tests/integration/Makefile) makes their own `#include "traps.h"` a no-op.
#ifndef TRAPS_H
#define TRAPS_H

typedef enum {st3bt, st1bt, st1bt_dct, stlcrc} system_type;

class Traps
{
	protected:
		void *ctx;              // lcrc_ctx of libphnrec_lcrc.so
		int nbanks;
		int trap_len;
		bool useHamming;
		bool add_c0;
		int bunchSize;
		system_type system;

	public:
		Traps();
		~Traps();
		void Init(char *dir);
		void CalcFeatures(float *band_energies, float *features, int n = 1, bool neededFea = true);
		void CalcFeaturesBunched(float *band_energies, float *features, int n = 1, bool neededFea = true);
		void Reset();
		int GetNumOuts();
		int GetTrapShift()        {return (trap_len - 1) / 2;};
		int GetDelay();
		// these functions should be called before Init
		bool SetSystem(char *sys);
		void SetTrapLen(int v)    {trap_len = v;};
		void SetHamming(bool ham) {useHamming = ham;};
		void SetNBanks(int v)     {nbanks = v;};
		void SetAddC0(bool v)     {add_c0 = v;};
		void SetBunchSize(int v)  {bunchSize = v;};
};

#endif
