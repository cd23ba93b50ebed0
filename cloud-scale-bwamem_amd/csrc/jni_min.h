// jni_min.h -- the part of the Java Native Interface ABI this library uses, declared from the JNI
// specification (function-table slot numbers are fixed by the spec).  The build image ships no JDK, so
// there is no <jni.h> to include; a JVM calling these entry points passes its real JNIEnv, whose function
// table has exactly this layout.
#pragma once

#include <stdarg.h>
#include <stdint.h>

extern "C" {

typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef _jobject* jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
typedef jarray jshortArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
struct _jfieldID;
typedef _jfieldID* jfieldID;
struct _jmethodID;
typedef _jmethodID* jmethodID;

#define JNI_OK 0
#define JNI_ABORT 2
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

struct JNINativeInterface_;
typedef const JNINativeInterface_* JNIEnv;  // C view: JNIEnv is a pointer to the function table

// Slot numbers from the JNI specification, "Interface Function Table".
enum {
  JNI_SLOT_FindClass = 6,
  JNI_SLOT_ThrowNew = 14,
  JNI_SLOT_ExceptionClear = 17,
  JNI_SLOT_PushLocalFrame = 19,
  JNI_SLOT_PopLocalFrame = 20,
  JNI_SLOT_NewGlobalRef = 21,
  JNI_SLOT_DeleteGlobalRef = 22,
  JNI_SLOT_DeleteLocalRef = 23,
  JNI_SLOT_AllocObject = 27,
  JNI_SLOT_GetMethodID = 33,
  JNI_SLOT_CallIntMethod = 49,
  JNI_SLOT_GetFieldID = 94,
  JNI_SLOT_GetObjectField = 95,
  JNI_SLOT_GetIntField = 100,
  JNI_SLOT_GetLongField = 101,
  JNI_SLOT_GetFloatField = 102,
  JNI_SLOT_GetDoubleField = 103,
  JNI_SLOT_SetObjectField = 104,
  JNI_SLOT_SetIntField = 109,
  JNI_SLOT_SetLongField = 110,
  JNI_SLOT_GetStaticMethodID = 113,
  JNI_SLOT_CallStaticObjectMethod = 114,
  JNI_SLOT_GetArrayLength = 171,
  JNI_SLOT_NewObjectArray = 172,
  JNI_SLOT_GetObjectArrayElement = 173,
  JNI_SLOT_SetObjectArrayElement = 174,
  JNI_SLOT_NewByteArray = 176,
  JNI_SLOT_NewShortArray = 178,
  JNI_SLOT_NewLongArray = 180,
  JNI_SLOT_GetByteArrayRegion = 200,
  JNI_SLOT_GetIntArrayRegion = 203,
  JNI_SLOT_GetLongArrayRegion = 204,
  JNI_SLOT_GetDoubleArrayRegion = 206,
  JNI_SLOT_SetByteArrayRegion = 208,
  JNI_SLOT_SetShortArrayRegion = 210,
  JNI_SLOT_SetLongArrayRegion = 212,
  JNI_SLOT_ExceptionCheck = 228,
  JNI_SLOT_COUNT = 233
};

struct JNINativeInterface_ {
  void* slot[JNI_SLOT_COUNT];
};

}  // extern "C"

// Typed accessors over the table (what the C++ JNIEnv_ wrapper of a real <jni.h> provides).
namespace jni {
template <class Fn>
inline Fn fn(JNIEnv* env, int slot) { return reinterpret_cast<Fn>((*env)->slot[slot]); }

inline jclass FindClass(JNIEnv* e, const char* n) { return fn<jclass (*)(JNIEnv*, const char*)>(e, JNI_SLOT_FindClass)(e, n); }
inline jint ThrowNew(JNIEnv* e, jclass c, const char* m) { return fn<jint (*)(JNIEnv*, jclass, const char*)>(e, JNI_SLOT_ThrowNew)(e, c, m); }
inline void ExceptionClear(JNIEnv* e) { fn<void (*)(JNIEnv*)>(e, JNI_SLOT_ExceptionClear)(e); }
inline jboolean ExceptionCheck(JNIEnv* e) { return fn<jboolean (*)(JNIEnv*)>(e, JNI_SLOT_ExceptionCheck)(e); }
inline jint PushLocalFrame(JNIEnv* e, jint cap) { return fn<jint (*)(JNIEnv*, jint)>(e, JNI_SLOT_PushLocalFrame)(e, cap); }
inline jobject PopLocalFrame(JNIEnv* e, jobject r) { return fn<jobject (*)(JNIEnv*, jobject)>(e, JNI_SLOT_PopLocalFrame)(e, r); }
inline jobject NewGlobalRef(JNIEnv* e, jobject o) { return fn<jobject (*)(JNIEnv*, jobject)>(e, JNI_SLOT_NewGlobalRef)(e, o); }
inline void DeleteGlobalRef(JNIEnv* e, jobject o) { fn<void (*)(JNIEnv*, jobject)>(e, JNI_SLOT_DeleteGlobalRef)(e, o); }
inline void DeleteLocalRef(JNIEnv* e, jobject o) { fn<void (*)(JNIEnv*, jobject)>(e, JNI_SLOT_DeleteLocalRef)(e, o); }
inline jobject AllocObject(JNIEnv* e, jclass c) { return fn<jobject (*)(JNIEnv*, jclass)>(e, JNI_SLOT_AllocObject)(e, c); }
inline jmethodID GetMethodID(JNIEnv* e, jclass c, const char* n, const char* s) { return fn<jmethodID (*)(JNIEnv*, jclass, const char*, const char*)>(e, JNI_SLOT_GetMethodID)(e, c, n, s); }
inline jint CallIntMethod(JNIEnv* e, jobject o, jmethodID m) { return fn<jint (*)(JNIEnv*, jobject, jmethodID, ...)>(e, JNI_SLOT_CallIntMethod)(e, o, m); }
inline jfieldID GetFieldID(JNIEnv* e, jclass c, const char* n, const char* s) { return fn<jfieldID (*)(JNIEnv*, jclass, const char*, const char*)>(e, JNI_SLOT_GetFieldID)(e, c, n, s); }
inline jobject GetObjectField(JNIEnv* e, jobject o, jfieldID f) { return fn<jobject (*)(JNIEnv*, jobject, jfieldID)>(e, JNI_SLOT_GetObjectField)(e, o, f); }
inline jint GetIntField(JNIEnv* e, jobject o, jfieldID f) { return fn<jint (*)(JNIEnv*, jobject, jfieldID)>(e, JNI_SLOT_GetIntField)(e, o, f); }
inline jlong GetLongField(JNIEnv* e, jobject o, jfieldID f) { return fn<jlong (*)(JNIEnv*, jobject, jfieldID)>(e, JNI_SLOT_GetLongField)(e, o, f); }
inline jfloat GetFloatField(JNIEnv* e, jobject o, jfieldID f) { return fn<jfloat (*)(JNIEnv*, jobject, jfieldID)>(e, JNI_SLOT_GetFloatField)(e, o, f); }
inline jdouble GetDoubleField(JNIEnv* e, jobject o, jfieldID f) { return fn<jdouble (*)(JNIEnv*, jobject, jfieldID)>(e, JNI_SLOT_GetDoubleField)(e, o, f); }
inline void SetObjectField(JNIEnv* e, jobject o, jfieldID f, jobject v) { fn<void (*)(JNIEnv*, jobject, jfieldID, jobject)>(e, JNI_SLOT_SetObjectField)(e, o, f, v); }
inline void SetIntField(JNIEnv* e, jobject o, jfieldID f, jint v) { fn<void (*)(JNIEnv*, jobject, jfieldID, jint)>(e, JNI_SLOT_SetIntField)(e, o, f, v); }
inline void SetLongField(JNIEnv* e, jobject o, jfieldID f, jlong v) { fn<void (*)(JNIEnv*, jobject, jfieldID, jlong)>(e, JNI_SLOT_SetLongField)(e, o, f, v); }
inline jmethodID GetStaticMethodID(JNIEnv* e, jclass c, const char* n, const char* s) { return fn<jmethodID (*)(JNIEnv*, jclass, const char*, const char*)>(e, JNI_SLOT_GetStaticMethodID)(e, c, n, s); }
inline jobject CallStaticObjectMethod(JNIEnv* e, jclass c, jmethodID m) { return fn<jobject (*)(JNIEnv*, jclass, jmethodID, ...)>(e, JNI_SLOT_CallStaticObjectMethod)(e, c, m); }
inline jsize GetArrayLength(JNIEnv* e, jarray a) { return fn<jsize (*)(JNIEnv*, jarray)>(e, JNI_SLOT_GetArrayLength)(e, a); }
inline jobjectArray NewObjectArray(JNIEnv* e, jsize n, jclass c, jobject init) { return fn<jobjectArray (*)(JNIEnv*, jsize, jclass, jobject)>(e, JNI_SLOT_NewObjectArray)(e, n, c, init); }
inline jobject GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) { return fn<jobject (*)(JNIEnv*, jobjectArray, jsize)>(e, JNI_SLOT_GetObjectArrayElement)(e, a, i); }
inline void SetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i, jobject v) { fn<void (*)(JNIEnv*, jobjectArray, jsize, jobject)>(e, JNI_SLOT_SetObjectArrayElement)(e, a, i, v); }
inline jshortArray NewShortArray(JNIEnv* e, jsize n) { return fn<jshortArray (*)(JNIEnv*, jsize)>(e, JNI_SLOT_NewShortArray)(e, n); }
inline jlongArray NewLongArray(JNIEnv* e, jsize n) { return fn<jlongArray (*)(JNIEnv*, jsize)>(e, JNI_SLOT_NewLongArray)(e, n); }
inline void SetLongArrayRegion(JNIEnv* e, jlongArray a, jsize s, jsize l, const jlong* b) { fn<void (*)(JNIEnv*, jlongArray, jsize, jsize, const jlong*)>(e, JNI_SLOT_SetLongArrayRegion)(e, a, s, l, b); }
inline void GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, jbyte* b) { fn<void (*)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*)>(e, JNI_SLOT_GetByteArrayRegion)(e, a, s, l, b); }
inline void GetIntArrayRegion(JNIEnv* e, jintArray a, jsize s, jsize l, jint* b) { fn<void (*)(JNIEnv*, jintArray, jsize, jsize, jint*)>(e, JNI_SLOT_GetIntArrayRegion)(e, a, s, l, b); }
inline void GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize s, jsize l, jlong* b) { fn<void (*)(JNIEnv*, jlongArray, jsize, jsize, jlong*)>(e, JNI_SLOT_GetLongArrayRegion)(e, a, s, l, b); }
inline jbyteArray NewByteArray(JNIEnv* e, jsize n) { return fn<jbyteArray (*)(JNIEnv*, jsize)>(e, JNI_SLOT_NewByteArray)(e, n); }
inline void SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, const jbyte* b) { fn<void (*)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*)>(e, JNI_SLOT_SetByteArrayRegion)(e, a, s, l, b); }
inline void GetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize s, jsize l, jdouble* b) { fn<void (*)(JNIEnv*, jdoubleArray, jsize, jsize, jdouble*)>(e, JNI_SLOT_GetDoubleArrayRegion)(e, a, s, l, b); }
inline void SetShortArrayRegion(JNIEnv* e, jshortArray a, jsize s, jsize l, const jshort* b) { fn<void (*)(JNIEnv*, jshortArray, jsize, jsize, const jshort*)>(e, JNI_SLOT_SetShortArrayRegion)(e, a, s, l, b); }
}  // namespace jni
