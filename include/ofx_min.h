/*
 * ofx_min.h -- the part of the OpenFX 1.x C API that the openfx-opencv plugins use.
 *
 * The reference builds against the `openfx/` submodule (include/ofxCore.h, ofxImageEffect.h, ofxProperty.h,
 * ofxParam.h, ofxMemory.h, ofxMultiThread.h, ofxMessage.h), which is an EMPTY directory in the reference tree
 * (.gitmodules:1-6).  These declarations are written out from the public OpenFX 1.x specification: struct
 * layouts, status codes and the property / action / suite strings must match the standard exactly, because a
 * host (Natron, Nuke, ...) binds to them by layout and by string.  Only what the three plugins touch is
 * declared (see SURVEY.md section 8(b) for the list and the reference call sites).
 */
#ifndef OFX_MIN_H
#define OFX_MIN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OfxExport __attribute__((visibility("default")))

typedef int OfxStatus;
typedef double OfxTime;
typedef struct OfxPropertySetStruct *OfxPropertySetHandle;
typedef struct OfxImageEffectStruct *OfxImageEffectHandle;
typedef struct OfxImageClipStruct *OfxImageClipHandle;
typedef struct OfxImageMemoryStruct *OfxImageMemoryHandle;
typedef struct OfxParamStruct *OfxParamHandle;
typedef struct OfxParamSetStruct *OfxParamSetHandle;
typedef struct OfxMutex *OfxMutexHandle;

typedef struct OfxRectI { int x1, y1, x2, y2; } OfxRectI;
typedef struct OfxRectD { double x1, y1, x2, y2; } OfxRectD;
typedef struct OfxPointD { double x, y; } OfxPointD;
typedef struct OfxRangeD { double min, max; } OfxRangeD;
typedef struct OfxRGBAColourB { unsigned char r, g, b, a; } OfxRGBAColourB;

/* status codes (ofxCore.h) */
#define kOfxStatOK 0
#define kOfxStatFailed 1
#define kOfxStatErrFatal 2
#define kOfxStatErrUnknown 3
#define kOfxStatErrMissingHostFeature 4
#define kOfxStatErrUnsupported 5
#define kOfxStatErrExists 6
#define kOfxStatErrFormat 7
#define kOfxStatErrMemory 8
#define kOfxStatErrBadHandle 9
#define kOfxStatErrBadIndex 10
#define kOfxStatErrValue 11
#define kOfxStatReplyYes 12
#define kOfxStatReplyNo 13
#define kOfxStatReplyDefault 14
#define kOfxStatErrImageFormat 1000

typedef struct OfxHost {
    OfxPropertySetHandle host;
    const void *(*fetchSuite)(OfxPropertySetHandle host, const char *suiteName, int suiteVersion);
} OfxHost;

typedef OfxStatus(OfxPluginEntryPoint)(const char *action, const void *handle, OfxPropertySetHandle inArgs, OfxPropertySetHandle outArgs);

typedef struct OfxPlugin {
    const char *pluginApi;
    int apiVersion;
    const char *pluginIdentifier;
    unsigned int pluginVersionMajor;
    unsigned int pluginVersionMinor;
    void (*setHost)(OfxHost *host);
    OfxPluginEntryPoint *mainEntry;
} OfxPlugin;

/* the two symbols a plugin binary exports */
OfxExport OfxPlugin *OfxGetPlugin(int nth);
OfxExport int OfxGetNumberOfPlugins(void);

/* ---- actions ---- */
#define kOfxImageEffectPluginApi "OfxImageEffectPluginAPI"
#define kOfxActionLoad "OfxActionLoad"
#define kOfxActionUnload "OfxActionUnload"
#define kOfxActionDescribe "OfxActionDescribe"
#define kOfxActionCreateInstance "OfxActionCreateInstance"
#define kOfxActionDestroyInstance "OfxActionDestroyInstance"
#define kOfxActionInstanceChanged "OfxActionInstanceChanged"
#define kOfxImageEffectActionDescribeInContext "OfxImageEffectActionDescribeInContext"
#define kOfxImageEffectActionRender "OfxImageEffectActionRender"
#define kOfxImageEffectActionGetFramesNeeded "OfxImageEffectActionGetFramesNeeded"

/* ---- suites ---- */
#define kOfxImageEffectSuite "OfxImageEffectSuite"
#define kOfxPropertySuite "OfxPropertySuite"
#define kOfxParameterSuite "OfxParameterSuite"
#define kOfxMemorySuite "OfxMemorySuite"
#define kOfxMultiThreadSuite "OfxMultiThreadSuite"
#define kOfxMessageSuite "OfxMessageSuite"
#define kOfxParamPropChoiceLabelOption "OfxParamPropChoiceLabelOption" /* per-option help text (host extension used by the Support library) */
#define kOfxInteractSuite "OfxInteractSuite"

/* ---- generic properties ---- */
#define kOfxPropTime "OfxPropTime"
#define kOfxPropName "OfxPropName"
#define kOfxPropType "OfxPropType"
#define kOfxPropLabel "OfxPropLabel"
#define kOfxPropShortLabel "OfxPropShortLabel"
#define kOfxPropLongLabel "OfxPropLongLabel"
#define kOfxPropPluginDescription "OfxPropPluginDescription"
#define kOfxPropInstanceData "OfxPropInstanceData"
#define kOfxPropChangeReason "OfxPropChangeReason"
#define kOfxChangeUserEdited "OfxChangeUserEdited"
#define kOfxTypeParameter "OfxTypeParameter"

/* ---- image effect properties ---- */
#define kOfxImageEffectPropContext "OfxImageEffectPropContext"
#define kOfxImageEffectContextFilter "OfxImageEffectContextFilter"
#define kOfxImageEffectContextGeneral "OfxImageEffectContextGeneral"
#define kOfxImageEffectPropSupportedContexts "OfxImageEffectPropSupportedContexts"
#define kOfxImageEffectPropSupportedPixelDepths "OfxImageEffectPropSupportedPixelDepths"
#define kOfxImageEffectPropSupportedComponents "OfxImageEffectPropSupportedComponents"
#define kOfxImageEffectPropSupportsMultipleClipDepths "OfxImageEffectPropMultipleClipDepths"
#define kOfxImageEffectPropSupportsMultipleClipPARs "OfxImageEffectPropSupportsMultipleClipPARs"
#define kOfxImageEffectPropSupportsMultiResolution "OfxImageEffectPropSupportsMultiResolution"
#define kOfxImageEffectPropSupportsTiles "OfxImageEffectPropSupportsTiles"
#define kOfxImageEffectPropTemporalClipAccess "OfxImageEffectPropTemporalClipAccess"
#define kOfxImageEffectPluginPropGrouping "OfxImageEffectPluginPropGrouping"
#define kOfxImageEffectPluginPropSingleInstance "OfxImageEffectPluginPropSingleInstance"
#define kOfxImageEffectPluginPropHostFrameThreading "OfxImageEffectPluginPropHostFrameThreading"
#define kOfxImageEffectPluginPropFieldRenderTwiceAlways "OfxImageEffectPluginPropFieldRenderTwiceAlways"
#define kOfxImageEffectPluginRenderThreadSafety "OfxImageEffectPluginRenderThreadSafety"
#define kOfxImageEffectRenderUnsafe "OfxImageEffectRenderUnsafe"
#define kOfxImageEffectRenderInstanceSafe "OfxImageEffectRenderInstanceSafe"
#define kOfxImageEffectRenderFullySafe "OfxImageEffectRenderFullySafe"
#define kOfxImageEffectPropRenderWindow "OfxImageEffectPropRenderWindow"
#define kOfxImageEffectPropRenderScale "OfxImageEffectPropRenderScale"
#define kOfxImageEffectPropFieldToRender "OfxImageEffectPropFieldToRender"
#define kOfxImageEffectPropPixelDepth "OfxImageEffectPropPixelDepth"
#define kOfxImageEffectPropComponents "OfxImageEffectPropComponents"
#define kOfxImageEffectPropFrameRange "OfxImageEffectPropFrameRange"
#define kOfxBitDepthNone "OfxBitDepthNone"
#define kOfxBitDepthByte "OfxBitDepthByte"
#define kOfxBitDepthShort "OfxBitDepthShort"
#define kOfxBitDepthFloat "OfxBitDepthFloat"
#define kOfxImageComponentNone "OfxImageComponentNone"
#define kOfxImageComponentRGBA "OfxImageComponentRGBA"
#define kOfxImageComponentRGB "OfxImageComponentRGB"
#define kOfxImageComponentAlpha "OfxImageComponentAlpha"
#define kOfxImageFieldNone "OfxFieldNone"

/* ---- clips and images ---- */
#define kOfxImageEffectOutputClipName "Output"
#define kOfxImageEffectSimpleSourceClipName "Source"
#define kOfxImageClipPropConnected "OfxImageClipPropConnected"
#define kOfxImageClipPropIsMask "OfxImageClipPropIsMask"
#define kOfxImageClipPropOptional "OfxImageClipPropOptional"
#define kOfxImagePropData "OfxImagePropData"
#define kOfxImagePropBounds "OfxImagePropBounds"
#define kOfxImagePropRowBytes "OfxImagePropRowBytes"
#define kOfxImagePropField "OfxImagePropField"
#define kOfxImagePropUniqueIdentifier "OfxImagePropUniqueIdentifier"  /* string: changes whenever the image's pixels change */
/* getFramesNeeded outArgs: "OfxImageClipPropFrameRange_" + clip name */
#define kOfxImageClipPropFrameRangePrefix "OfxImageClipPropFrameRange_"

/* ---- parameters ---- */
#define kOfxParamTypeInteger "OfxParamTypeInteger"
#define kOfxParamTypeDouble "OfxParamTypeDouble"
#define kOfxParamTypeChoice "OfxParamTypeChoice"
#define kOfxParamTypePage "OfxParamTypePage"
#define kOfxParamPropDefault "OfxParamPropDefault"
#define kOfxParamPropMin "OfxParamPropMin"
#define kOfxParamPropMax "OfxParamPropMax"
#define kOfxParamPropDisplayMin "OfxParamPropDisplayMin"
#define kOfxParamPropDisplayMax "OfxParamPropDisplayMax"
#define kOfxParamPropHint "OfxParamPropHint"
#define kOfxParamPropScriptName "OfxParamPropScriptName"
#define kOfxParamPropDoubleType "OfxParamPropDoubleType"
#define kOfxParamDoubleTypeScale "OfxParamDoubleTypeScale"
#define kOfxParamDoubleTypePlain "OfxParamDoubleTypePlain"
#define kOfxParamPropAnimates "OfxParamPropAnimates"
#define kOfxParamPropSecret "OfxParamPropSecret"
#define kOfxParamPropChoiceOption "OfxParamPropChoiceOption"
#define kOfxParamPropPageChild "OfxParamPropPageChild"
#define kOfxParamPropIncrement "OfxParamPropIncrement"
#define kOfxParamPropDigits "OfxParamPropDigits"

/* ---- suite structs (function order is ABI) ---- */
typedef struct OfxPropertySuiteV1 {
    OfxStatus (*propSetPointer)(OfxPropertySetHandle properties, const char *property, int index, void *value);
    OfxStatus (*propSetString)(OfxPropertySetHandle properties, const char *property, int index, const char *value);
    OfxStatus (*propSetDouble)(OfxPropertySetHandle properties, const char *property, int index, double value);
    OfxStatus (*propSetInt)(OfxPropertySetHandle properties, const char *property, int index, int value);
    OfxStatus (*propSetPointerN)(OfxPropertySetHandle properties, const char *property, int count, void *const *value);
    OfxStatus (*propSetStringN)(OfxPropertySetHandle properties, const char *property, int count, const char *const *value);
    OfxStatus (*propSetDoubleN)(OfxPropertySetHandle properties, const char *property, int count, const double *value);
    OfxStatus (*propSetIntN)(OfxPropertySetHandle properties, const char *property, int count, const int *value);
    OfxStatus (*propGetPointer)(OfxPropertySetHandle properties, const char *property, int index, void **value);
    OfxStatus (*propGetString)(OfxPropertySetHandle properties, const char *property, int index, char **value);
    OfxStatus (*propGetDouble)(OfxPropertySetHandle properties, const char *property, int index, double *value);
    OfxStatus (*propGetInt)(OfxPropertySetHandle properties, const char *property, int index, int *value);
    OfxStatus (*propGetPointerN)(OfxPropertySetHandle properties, const char *property, int count, void **value);
    OfxStatus (*propGetStringN)(OfxPropertySetHandle properties, const char *property, int count, char **value);
    OfxStatus (*propGetDoubleN)(OfxPropertySetHandle properties, const char *property, int count, double *value);
    OfxStatus (*propGetIntN)(OfxPropertySetHandle properties, const char *property, int count, int *value);
    OfxStatus (*propReset)(OfxPropertySetHandle properties, const char *property);
    OfxStatus (*propGetDimension)(OfxPropertySetHandle properties, const char *property, int *count);
} OfxPropertySuiteV1;

typedef struct OfxImageEffectSuiteV1 {
    OfxStatus (*getPropertySet)(OfxImageEffectHandle imageEffect, OfxPropertySetHandle *propHandle);
    OfxStatus (*getParamSet)(OfxImageEffectHandle imageEffect, OfxParamSetHandle *paramSet);
    OfxStatus (*clipDefine)(OfxImageEffectHandle imageEffect, const char *name, OfxPropertySetHandle *propertySet);
    OfxStatus (*clipGetHandle)(OfxImageEffectHandle imageEffect, const char *name, OfxImageClipHandle *clip, OfxPropertySetHandle *propertySet);
    OfxStatus (*clipGetPropertySet)(OfxImageClipHandle clip, OfxPropertySetHandle *propHandle);
    OfxStatus (*clipGetImage)(OfxImageClipHandle clip, OfxTime time, const OfxRectD *region, OfxPropertySetHandle *imageHandle);
    OfxStatus (*clipReleaseImage)(OfxPropertySetHandle imageHandle);
    OfxStatus (*clipGetRegionOfDefinition)(OfxImageClipHandle clip, OfxTime time, OfxRectD *bounds);
    int (*abort)(OfxImageEffectHandle imageEffect);
    OfxStatus (*imageMemoryAlloc)(OfxImageEffectHandle instanceHandle, size_t nBytes, OfxImageMemoryHandle *memoryHandle);
    OfxStatus (*imageMemoryFree)(OfxImageMemoryHandle memoryHandle);
    OfxStatus (*imageMemoryLock)(OfxImageMemoryHandle memoryHandle, void **returnedPtr);
    OfxStatus (*imageMemoryUnlock)(OfxImageMemoryHandle memoryHandle);
} OfxImageEffectSuiteV1;

typedef struct OfxParameterSuiteV1 {
    OfxStatus (*paramDefine)(OfxParamSetHandle paramSet, const char *paramType, const char *name, OfxPropertySetHandle *propertySet);
    OfxStatus (*paramGetHandle)(OfxParamSetHandle paramSet, const char *name, OfxParamHandle *param, OfxPropertySetHandle *propertySet);
    OfxStatus (*paramSetGetPropertySet)(OfxParamSetHandle paramSet, OfxPropertySetHandle *propHandle);
    OfxStatus (*paramGetPropertySet)(OfxParamHandle param, OfxPropertySetHandle *propHandle);
    OfxStatus (*paramGetValue)(OfxParamHandle paramHandle, ...);
    OfxStatus (*paramGetValueAtTime)(OfxParamHandle paramHandle, OfxTime time, ...);
    OfxStatus (*paramGetDerivative)(OfxParamHandle paramHandle, OfxTime time, ...);
    OfxStatus (*paramGetIntegral)(OfxParamHandle paramHandle, OfxTime time1, OfxTime time2, ...);
    OfxStatus (*paramSetValue)(OfxParamHandle paramHandle, ...);
    OfxStatus (*paramSetValueAtTime)(OfxParamHandle paramHandle, OfxTime time, ...);
    OfxStatus (*paramGetNumKeys)(OfxParamHandle paramHandle, unsigned int *numberOfKeys);
    OfxStatus (*paramGetKeyTime)(OfxParamHandle paramHandle, unsigned int nthKey, OfxTime *time);
    OfxStatus (*paramGetKeyIndex)(OfxParamHandle paramHandle, OfxTime time, int direction, int *index);
    OfxStatus (*paramDeleteKey)(OfxParamHandle paramHandle, OfxTime time);
    OfxStatus (*paramDeleteAllKeys)(OfxParamHandle paramHandle);
    OfxStatus (*paramCopy)(OfxParamHandle paramTo, OfxParamHandle paramFrom, OfxTime dstOffset, const OfxRangeD *frameRange);
    OfxStatus (*paramEditBegin)(OfxParamSetHandle paramSet, const char *name);
    OfxStatus (*paramEditEnd)(OfxParamSetHandle paramSet);
} OfxParameterSuiteV1;

typedef struct OfxMemorySuiteV1 {
    OfxStatus (*memoryAlloc)(void *handle, size_t nBytes, void **allocatedData);
    OfxStatus (*memoryFree)(void *allocatedData);
} OfxMemorySuiteV1;

typedef void(OfxThreadFunctionV1)(unsigned int threadIndex, unsigned int threadMax, void *customArg);
typedef struct OfxMultiThreadSuiteV1 {
    OfxStatus (*multiThread)(OfxThreadFunctionV1 func, unsigned int nThreads, void *customArg);
    OfxStatus (*multiThreadNumCPUs)(unsigned int *nCPUs);
    OfxStatus (*multiThreadIndex)(unsigned int *threadIndex);
    int (*multiThreadIsSpawnedThread)(void);
    OfxStatus (*mutexCreate)(OfxMutexHandle *mutex, int lockCount);
    OfxStatus (*mutexDestroy)(const OfxMutexHandle mutex);
    OfxStatus (*mutexLock)(const OfxMutexHandle mutex);
    OfxStatus (*mutexUnLock)(const OfxMutexHandle mutex);
    OfxStatus (*mutexTryLock)(const OfxMutexHandle mutex);
} OfxMultiThreadSuiteV1;

#define kOfxMessageError "OfxMessageError"
#define kOfxMessageLog "OfxMessageLog"
#define kOfxMessageMessage "OfxMessageMessage"
typedef struct OfxMessageSuiteV1 {
    OfxStatus (*message)(void *handle, const char *messageType, const char *messageId, const char *format, ...);
} OfxMessageSuiteV1;

#ifdef __cplusplus
}
#endif
#endif
